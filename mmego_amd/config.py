"""Run-time configuration singletons, mutated by main.py's argparse exactly like the reference's
Config/config.py:11-70 and Config/config_demo.py:11-60 (class attributes used as flags; CLI > defaults).
"""
import os

import numpy as np
import torch

from . import skeleton as sk

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _default_device():
    return "cuda:0" if torch.cuda.is_available() else "cpu"


def _pretrained(*parts):
    root = os.environ.get("MMEGO_PRETRAINED_ROOT", os.path.join(_REPO, "Resource", "Pretrained_model"))
    return os.path.join(root, *parts)


class _Common:
    pb = 10
    colab = False
    frame_no = sk.FRAMES
    pc_no = sk.POINTS
    lower_pc_no = sk.LOWER_POINTS
    joint_num_all, joint_num_upper, joint_num_lower = sk.JOINTS_ALL, sk.JOINTS_UPPER, sk.JOINTS_LOWER
    num_action = 13
    IMU_used = True
    IMU_pretrained = Upper_pretrained = Lower_pretrained = False
    device = _default_device()
    skeleton_all = np.asarray(sk.BONES_ALL)
    skeleton_upper_body = np.asarray(sk.BONES_UPPER)
    skeleton_lower_body = np.asarray(sk.BONES_LOWER)
    kinect_upper_gragh = list(sk.GCN_EDGES)          # (sic) the reference's spelling
    kinect_joint_selection = list(sk.KINECT_SELECTION)
    upper_joint_map, lower_joint_map, hand_joint_map = list(sk.UPPER_MAP), list(sk.LOWER_MAP), list(sk.HAND_MAP)
    model_IMU_path = _pretrained("IMU_Net", "epoch173_batch20frame20lr3e-05.pth")
    model_upper_path = _pretrained("Upper_Net", "epoch451_batch20frame20lr3e-05.pth")
    model_lower_path = _pretrained("Lower_Net", "epoch161_batch20frame20lr0.0003.pth")
    dataset_random_seed = 1
    data_root = os.environ.get("MMEGO_DATA_ROOT", os.path.join(_REPO, "Resource", "Sample_data"))
    # Not in the reference: the shipped snapshot lacks the IMU_Net checkpoint (.MISSING_LARGE_BLOBS), so the head pose
    # can be taken from the recording instead (R = loader R_R0R, t = ground-truth head joint).
    gt_head_pose = False
    data_parallel = False       # set by main.py when launched under torch.distributed.run


class Config(_Common):
    Idx = 1001
    epochs = 600
    lr = 3e-5
    batch_size = 20


class ConfigDemo(_Common):
    Idx = 1
    batch_per_action = 3
