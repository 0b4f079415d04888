"""Skeleton topology constants of the path and the ST-GCN adjacency (host-side, numpy).

Mirrors the values of the reference's Config/config.py:22-55 and the 'kinect_upper' graph of
Net/GCN.py:150-278.  The forward-kinematics walk orders live in csrc/geom.hip (make_plan).
"""
import numpy as np

JOINTS_ALL, JOINTS_UPPER, JOINTS_LOWER = 21, 15, 8
POINTS = 128
LOWER_POINTS = 64
FRAMES = 20

BONES_UPPER = [[20, 3], [3, 2], [2, 1], [2, 4], [2, 8], [4, 5], [5, 6], [6, 7], [8, 9], [9, 10], [10, 11],
               [1, 0], [0, 12], [0, 16]]
BONES_LOWER = [[12, 13], [13, 14], [14, 15], [16, 17], [17, 18], [18, 19]]
BONES_ALL = BONES_UPPER + BONES_LOWER
UPPER_MAP = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 16, 20]
LOWER_MAP = [12, 13, 14, 15, 16, 17, 18, 19]
HAND_MAP = [7, 6, 11, 10]
KINECT_SELECTION = [0, 1, 2, 3, 4, 5, 6, 7, 11, 12, 13, 14, 18, 19, 20, 21, 22, 23, 24, 25, 26]
GCN_EDGES = [(0, 12), (0, 13), (0, 1), (1, 2), (2, 3), (2, 4), (2, 8), (3, 14), (4, 5), (5, 6), (6, 7), (8, 9),
             (9, 10), (10, 11)]


def gcn_adjacency(strategy="distance", max_hop=1):
    """(K, 15, 15) float64 partitioned normalised adjacency: K=1 ('uniform') or max_hop+1 ('distance')."""
    n = JOINTS_UPPER
    link = np.eye(n)
    for i, j in GCN_EDGES:
        link[i, j] = link[j, i] = 1.0
    # hop distance up to max_hop (inf beyond)
    hop = np.full((n, n), np.inf)
    power = [np.linalg.matrix_power(link, d) > 0 for d in range(max_hop + 1)]
    for d in range(max_hop, -1, -1):
        hop[power[d]] = d
    reach = (hop <= max_hop).astype(np.float64)
    deg = reach.sum(axis=0)
    inv_sqrt = np.where(deg > 0, deg ** -0.5, 0.0)
    norm = (inv_sqrt[:, None] * reach) * inv_sqrt[None, :]
    if strategy == "uniform":
        return norm[None].copy()
    if strategy == "distance":
        return np.stack([np.where(hop == h, norm, 0.0) for h in range(max_hop + 1)])
    raise ValueError("unsupported graph strategy %r" % (strategy,))
