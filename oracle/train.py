"""Loss, optimiser step and train_once bodies (oracle; test infrastructure only).

Restates Processor/Train/Train_Upper.py:134-187, Train_Lower.py:155-230 and
Train_IMU.py:21-34,114-149 as functions over explicit tensors (no DataLoader,
no logging).  Also the CPU-baseline timing harness used by bench.py.
"""
import math
import time

import torch

from . import skeleton as sk


def l1_sum(pred, target):
    """torch.nn.L1Loss(reduction='sum').  Train_Upper.py:53,179."""
    return (pred - target).abs().sum()


def geodesic_sum_deg(R, R_gt, eps=1e-7):
    """sum acos(clamp((tr(R R_gt^T)-1)/2)) * 180 / 3.14159265358.  Train_IMU.py:21-34,138."""
    m = torch.bmm(R.reshape(-1, 3, 3), R_gt.reshape(-1, 3, 3).transpose(1, 2))
    cos = (m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2] - 1) / 2
    return torch.acos(torch.clamp(cos, -1 + eps, 1 - eps)).sum() / 3.14159265358 * 180


def imu_loss(R, t, R_gt, head_gt):
    """Train_IMU.py:138-141."""
    return geodesic_sum_deg(R, R_gt) + 100 * torch.sqrt(((t - head_gt) ** 2).sum(-1)).sum()


def adam_update_(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """One torch.optim.Adam update written out (coupled L2 decay, no amsgrad).

    The reference calls torch.optim.Adam (Train_Upper.py:60); this is the
    published update rule of that optimiser in the operation order of
    torch/optim/adam.py `_single_tensor_adam`, used to pin the fused HIP Adam.
    """
    if weight_decay != 0.0:
        g = g + weight_decay * p
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def zeros_state(B, device="cpu"):
    return (torch.zeros(6, B, 64, device=device), torch.zeros(6, B, 64, device=device))


def upper_train_step(upper, imu_net, opt, x, imu, body, target):
    """One Train_Upper.train_once minibatch body.  Returns (loss, joints)."""
    B = x.shape[0]
    opt.zero_grad()
    h0, c0 = zeros_state(B)
    with torch.no_grad():
        R, t = imu_net(imu)
    joints, _, _, _, _ = upper(x, h0, c0, body, R, t)
    loss = l1_sum(joints, target[:, :, sk.UPPER_MAP, :])
    loss.backward()
    opt.step()
    return loss.detach(), joints.detach()


def lower_train_step(lower, upper, imu_net, opt, x, imu, body, target):
    """One Train_Lower.train_once minibatch body (frozen IMU and Upper forward inside)."""
    B = x.shape[0]
    opt.zero_grad()
    h0, c0 = zeros_state(B)
    with torch.no_grad():
        R, t = imu_net(imu)
        up, _, _, _, _ = upper(x, h0, c0, body, R, t)          # mutates x (Q1)
    joints, _ = lower(up.clone(), x, h0, c0, h0, c0, body, R, t)
    loss = l1_sum(joints, target[:, :, sk.LOWER_MAP, :])
    loss.backward()
    opt.step()
    return loss.detach(), joints.detach()


def time_ul_step(upper, lower, imu_net, x, imu, body, target, steps=3, warmup=1, lr=3e-5):
    """Wall time of the metric unit "U+L step" on CPU: returns (t_upper_s, t_lower_s)."""
    from copy import deepcopy
    up_frozen = deepcopy(upper).eval()
    imu_net.eval()
    upper.train()
    lower.train()
    opt_u = torch.optim.Adam(upper.parameters(), lr=lr)
    opt_l = torch.optim.Adam(lower.parameters(), lr=lr)
    tu = tl = 0.0
    for it in range(warmup + steps):
        t0 = time.perf_counter()
        upper_train_step(upper, imu_net, opt_u, x.clone(), imu, body, target)
        t1 = time.perf_counter()
        lower_train_step(lower, up_frozen, imu_net, opt_l, x.clone(), imu, body, target)
        t2 = time.perf_counter()
        if it >= warmup:
            tu += t1 - t0
            tl += t2 - t1
    return tu / steps, tl / steps
