"""Joint-error metric of ``--infer`` (oracle; test infrastructure only).

Restates Processor/Test/Demo_test.py:64-69,121-123,150-180.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import skeleton as sk


def assemble(upper_l, lower_l):
    """(B,T,15,3),(B,T,8,3) -> (B,T,21,3); lower overwrites the shared hips.  Demo_test.py:121-123."""
    B, T = upper_l.shape[:2]
    full = torch.zeros(B, T, sk.JOINTS_ALL, 3, dtype=upper_l.dtype)
    full[:, :, list(sk.UPPER_MAP)] = upper_l
    full[:, :, list(sk.LOWER_MAP)] = lower_l
    return full


def bone_angle_deg(pred, target):
    """(B,T,21,3) x2 -> (B,T,20) degrees.  Demo_test.py:64-69."""
    root = [p for p, _ in sk.BONES_ALL]
    leaf = [c for _, c in sk.BONES_ALL]
    cs = F.cosine_similarity(pred[:, :, leaf] - pred[:, :, root], target[:, :, leaf] - target[:, :, root], dim=-1)
    return torch.abs(torch.acos(torch.clamp(cs, -1.0, 1.0)) / 3.14159265358 * 180.0)


def batch_errors(upper_l, lower_l, target):
    """Per-batch figures that Demo_test.eval_model appends to its lists."""
    pred = assemble(upper_l, lower_l)
    dist = torch.sqrt(((pred - target) ** 2).sum(-1))
    up = torch.sqrt(((upper_l - target[:, :, list(sk.UPPER_MAP)]) ** 2).sum(-1)).mean().item()
    lo = torch.sqrt(((lower_l - target[:, :, list(sk.LOWER_MAP)]) ** 2).sum(-1)).mean().item()
    per_joint = dist.mean(0).mean(0).numpy().tolist()
    angle = bone_angle_deg(pred, target).mean(0).mean(0).numpy().tolist()
    return dist.mean().item(), up, lo, per_joint, angle


def summarize(per_batch):
    """Mean of per-batch means, x100 -> cm.  Demo_test.py:165-180."""
    all_, up, lo, pj, ang = zip(*per_batch)
    ang = np.mean(ang, axis=0)
    return {"all_cm": float(np.mean(all_) * 100), "upper_cm": float(np.mean(up) * 100),
            "lower_cm": float(np.mean(lo) * 100), "rot_deg": float(sum(ang) / len(ang)),
            "per_joint_cm": (np.mean(pj, axis=0) * 100).tolist()}
