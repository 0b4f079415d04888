"""Helpers the reference keeps in Util/Universal_Util/Utils.py: the two frame transforms (GPU, HIP kernels),
EarlyStopping and angle_minus.  Plotting helpers of that file are out of scope."""
import torch

from . import ops
from .config import Config


class EarlyStopping:
    """Stop when the validation loss has not improved for ``patience`` epochs (Utils.py:14-47)."""

    def __init__(self, patience=5, delta=0, verbose=False):
        self.patience, self.delta, self.verbose = patience, delta, verbose
        self.counter, self.best_score, self.early_stop = 0, None, False

    def __call__(self, val_loss):
        if self.best_score is None or val_loss <= self.best_score + self.delta:
            self.best_score, self.counter = val_loss, 0
        else:
            self.counter += 1
            if self.verbose:
                print(f"Validation loss increased [{self.counter}/{self.patience}]")
            self.early_stop = self.counter >= self.patience
        return self.early_stop


def Transform2H(points, batch_size, length_size, pc_no, R, t):
    """xyz <- R (xyz - t) per frame, IN PLACE on ``points`` (Utils.py:284-292); returns a (B*T, pc_no, C) view."""
    if not points.is_cuda:
        raise RuntimeError("Transform2H runs on the MI355X HIP path only; there is no CPU fallback")
    v = points.view(batch_size * length_size, pc_no, -1)
    if Config.IMU_used:
        ops.transform2h_(v, R.contiguous(), t.contiguous())
    return v


def Transform2R(points, batch_size, length_size, pc_no, R, t):
    """p <- R^T p + t per frame (Utils.py:274-281); returns (B, T, pc_no, 3)."""
    if not points.is_cuda:
        raise RuntimeError("Transform2R runs on the MI355X HIP path only; there is no CPU fallback")
    src = points.contiguous().view(batch_size * length_size, pc_no, 3)
    if not Config.IMU_used:
        return src.view(batch_size, length_size, pc_no, 3)
    out = torch.empty_like(src)
    ops.rotate_points(src, out, R.contiguous(), t.contiguous(), transpose=True)
    return out.view(batch_size, length_size, pc_no, 3)


def angle_minus(m1, m2):
    """Geodesic angle (deg) between rotation batches (Utils.py:263-271); small host-side helper."""
    m = torch.bmm(m1.view(-1, 3, 3), m2.view(-1, 3, 3).transpose(1, 2))
    cos = (m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2] - 1) / 2
    return torch.acos(torch.clamp(cos, -1 + 1e-7, 1 - 1e-7)) / 3.14159265358 * 180
