"""Skeleton-graph adjacency for the ST-GCN (oracle; test infrastructure only).

Restates Net/GCN.py:150-278 for the only layout used ('kinect_upper', 15
nodes, max_hop 1, dilation 1).  numpy float64, as the reference.
"""
import numpy as np

from . import skeleton as sk


def hop_distance(num_node, edges, max_hop=1):
    """GCN.py:242-255: shortest hop count up to max_hop, inf beyond."""
    adj = np.zeros((num_node, num_node))
    for i, j in edges:
        adj[i, j] = 1
        adj[j, i] = 1
    hop = np.full((num_node, num_node), np.inf)
    reach = [np.linalg.matrix_power(adj, d) > 0 for d in range(max_hop + 1)]
    for d in range(max_hop, -1, -1):
        hop[reach[d]] = d
    return hop


def normalize_undirected(adj):
    """D^-1/2 A D^-1/2 with D = column sums.  GCN.py:269-278."""
    deg = adj.sum(0)
    scale = np.zeros_like(deg)
    nz = deg > 0
    scale[nz] = deg[nz] ** -0.5
    return (scale[:, None] * adj) * scale[None, :]      # (Dn A) Dn, the reference's product order


def adjacency(strategy="distance", max_hop=1):
    """(K,15,15) float64.  'uniform' -> K=1, 'distance' -> K=max_hop+1.

    GCN.py:198-214.
    """
    n = sk.JOINTS_UPPER
    edges = [(i, i) for i in range(n)] + list(sk.GCN_EDGES)
    hop = hop_distance(n, edges, max_hop)
    reach = np.zeros((n, n))
    for h in range(max_hop + 1):
        reach[hop == h] = 1
    norm = normalize_undirected(reach)
    if strategy == "uniform":
        return norm[None].copy()
    if strategy == "distance":
        out = np.zeros((max_hop + 1, n, n))
        for h in range(max_hop + 1):
            out[h][hop == h] = norm[hop == h]
        return out
    raise ValueError("strategy must be 'uniform' or 'distance'")
