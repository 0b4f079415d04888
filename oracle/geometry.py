"""Geometry helpers of the hot path (oracle; test infrastructure only).

Plain torch, fp32, CPU.  Each function cites the reference lines it restates.
"""
import torch
import torch.nn.functional as F

from . import skeleton as sk


def transform_to_head_(points, R, t):
    """xyz <- R (xyz - t) per frame, IN PLACE on ``points`` (quirk Q1).

    points: (B, T, P, C>=3) or (B*T, P, C); R: (B, T, 3, 3); t: (B, T, 3).
    Reference Util/Universal_Util/Utils.py:284-292 (writes into a view of the
    caller's tensor).  Returns a (B*T, P, C) view of the same storage.
    """
    BT = R.shape[0] * R.shape[1]
    P = points.shape[-2]
    v = points.view(BT, P, -1)
    xyz = v[:, :, :3].clone()
    Rf = R.reshape(BT, 1, 3, 3)
    tf = t.reshape(BT, 1, 3)
    v[:, :, :3] = torch.matmul(Rf, (xyz - tf).unsqueeze(-1)).squeeze(-1)
    return v


def transform_to_world(points, R, t):
    """p <- R^T p + t per frame (new tensor).  Utils.py:274-281."""
    B, T = R.shape[0], R.shape[1]
    P = points.shape[-2]
    v = points.reshape(B * T, P, 3)
    Rt = R.reshape(B * T, 1, 3, 3).transpose(-1, -2)
    out = torch.matmul(Rt, v.unsqueeze(-1)).squeeze(-1) + t.reshape(B * T, 1, 3)
    return out.view(B, T, P, 3)


def rot6d_normalize(six):
    """(n,6) -> (n,3,3) Gram-Schmidt with F.normalize (eps 1e-12).

    Upper_Net.py:354-362 / Lower_Net.py:125-133: x=norm(a), z=norm(x X b),
    y = z X x, columns [x y z].
    """
    a, b = six[:, :3], six[:, 3:]
    x = F.normalize(a, dim=-1)
    z = F.normalize(torch.cross(x, b, dim=-1), dim=-1)
    y = torch.cross(z, x, dim=-1)
    return torch.stack((x, y, z), dim=-1)


def rot6d_imu(six):
    """Same construction with the IMU net's eps rule: v / max(|v|, 1e-8).

    IMU_Net.py:7-47.
    """
    def nrm(v):
        mag = torch.sqrt((v * v).sum(1, keepdim=True))
        return v / torch.clamp(mag, min=1e-8)

    a, b = six[:, :3], six[:, 3:]
    x = nrm(a)
    z = nrm(torch.cross(x, b, dim=-1))
    y = torch.cross(z, x, dim=-1)
    return torch.stack((x, y, z), dim=-1)


def _body_rows(initial_body, T):
    """Row n of the (B*T) frame list uses body of sample n % B (quirk Q2).

    Upper_Net.py:134 / Lower_Net.py:26: ``initial_body.view(B,M,3,1).repeat(L,1,1,1)``.
    """
    return initial_body.repeat(T, 1, 1)          # (T*B, M, 3): row n -> sample n % B


def fk_upper(q, initial_body, head):
    """Upper forward kinematics.  Upper_Net.py:122-144 (quirks Q2, Q4).

    q: (B,T,14,3,3), initial_body: (B,20,3), head: (B,T,3) -> (B,T,15,3).
    """
    B, T = q.shape[0], q.shape[1]
    qf = q.reshape(B * T, 14, 3, 3)
    body = _body_rows(initial_body, T)
    slots = [None] * sk.JOINTS_UPPER
    slots[sk.JOINTS_UPPER - 1] = head.reshape(B * T, 3)
    for parent, child, row in sk.upper_fk_plan():
        slots[child] = slots[parent] + torch.matmul(qf[:, child], body[:, row].unsqueeze(-1)).squeeze(-1)
    return torch.stack(slots, dim=1).view(B, T, sk.JOINTS_UPPER, 3)


def fk_lower(q, hip_l, hip_r, initial_body):
    """Lower forward kinematics.  Lower_Net.py:12-37.

    q: (B,T,6,3,3), hips: (B,T,3), initial_body: (B,20,3) -> (B,T,8,3).
    """
    B, T = q.shape[0], q.shape[1]
    qf = q.reshape(B * T, 6, 3, 3)
    body = _body_rows(initial_body, T)
    slots = [None] * sk.JOINTS_LOWER
    slots[0] = hip_l.reshape(B * T, 3)
    slots[4] = hip_r.reshape(B * T, 3)
    for parent, child, rot, row in sk.lower_fk_plan():
        slots[child] = slots[parent] + torch.matmul(qf[:, rot], body[:, row].unsqueeze(-1)).squeeze(-1)
    return torch.stack(slots, dim=1).view(B, T, sk.JOINTS_LOWER, 3)


# ----------------------------------------------------------------------------
# anchor ("voxel") grouping -- Upper_Net.py:10-119
# ----------------------------------------------------------------------------

def anchor_grid():
    """(27,3) anchors, layout [z][y][x]; x in {0,.3,.6}, y,z in {-.3,0,.3}.

    Upper_Net.py:75-97 (values built as min + n*interval in Python floats, then
    stored to fp32).
    """
    pts = []
    for zi in range(3):
        for yi in range(3):
            for xi in range(3):
                pts.append((0 + xi * 0.3, -0.3 + yi * 0.3, -0.3 + zi * 0.3))
    return torch.tensor(pts, dtype=torch.float32)


def square_distance(src, dst):
    """-2 src.dst^T + |src|^2 + |dst|^2, +inf where dst xyz == 0 (quirk Q5).

    Upper_Net.py:10-32.  src (B,S,3), dst (B,N,3) -> (B,S,N).
    """
    d = -2 * torch.matmul(src, dst.transpose(1, 2))
    d = d + (src ** 2).sum(-1).unsqueeze(-1)
    d = d + (dst ** 2).sum(-1).unsqueeze(1)
    dead = (dst == 0).all(dim=-1).unsqueeze(1).expand_as(d)
    return torch.where(dead, torch.full_like(d, float("inf")), d)


def group_indices(xyz, anchors, nsample=8):
    """int64 (B,S,nsample): the nsample nearest points per anchor.

    Upper_Net.py:54-72: full ascending sort of each distance row, first 8.
    Ties are broken lowest-index-first (stable sort) -- checked against the
    reference's output in tests/golden/g2_grouping.npz.
    """
    d = square_distance(anchors, xyz)
    order = torch.sort(d, dim=-1, stable=True).indices
    return order[:, :, :nsample]


def anchor_grouping(xyz, feats, nsample=8):
    """cat(anchor, xyz - anchor, feats) gathered per anchor.

    Upper_Net.py:100-119.  xyz (B,N,3), feats (B,N,D) -> (B,27,nsample,6+D), idx.
    """
    Bn = xyz.shape[0]
    anchors = anchor_grid().to(xyz.device).unsqueeze(0).expand(Bn, -1, -1)
    idx = group_indices(xyz, anchors, nsample)
    bi = torch.arange(Bn).view(Bn, 1, 1)
    g_xyz = xyz[bi, idx]
    g_feat = feats[bi, idx]
    a = anchors.unsqueeze(2).expand(-1, -1, nsample, -1)
    return torch.cat((a, g_xyz - a, g_feat), dim=-1), idx


def top_x_select(pts, keep=sk.LOWER_POINTS):
    """Keep the ``keep`` rows with the largest x (col 0), descending.

    Lower_Net.py:216-227.  pts (BT,N,C) -> (BT,keep,C), idx int64 (BT,keep).
    """
    order = torch.sort(pts[:, :, 0], dim=1, descending=True, stable=True).indices[:, :keep]
    bi = torch.arange(pts.shape[0]).view(-1, 1)
    return pts[bi, order], order
