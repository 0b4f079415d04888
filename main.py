#!/usr/bin/env python3
"""Command line of the reference (main.py:7-73), same flags and precedence (CLI > Config defaults):

  python main.py --train --network {IMU_Net,Upper_Net,Lower_Net} [--epochs N --lr F --batch_size N --device cuda:0
                 --log_dir IDX --load_IMU_path P --load_Upper_path P --load_Lower_path P]
  python main.py --infer [--vis]

Added flags (not in the reference): --gt_head_pose (use the recorded head pose when no IMU_Net checkpoint is
available), --data_root, --seed, --resume (bit-exact continuation: weights, Adam moments/step, epoch, RNG states).  Under `python -m torch.distributed.run --nproc-per-node N main.py --train ...` training is
data parallel (one rank per GPU, RCCL gradient all-reduce).
"""
import argparse
import os

import torch

from mmego_amd.config import Config, ConfigDemo


def build_parser():
    p = argparse.ArgumentParser(description="Processor collection")
    p.add_argument("--network", type=str, choices=["IMU_Net", "Upper_Net", "Lower_Net"],
                   help="Choose a network: IMU_Net, Upper_Net, Lower_Net")
    p.add_argument("--train", action="store_true", help="Train model")
    p.add_argument("--infer", action="store_true", help="Perform inference")
    p.add_argument("--vis", action="store_true", help="Visualization")
    p.add_argument("--colab", action="store_true", help="Called by colab")
    p.add_argument("--epochs", type=int, help="Number of epochs")
    p.add_argument("--lr", type=float, help="Learning rate")
    p.add_argument("--device", type=str, help="device: [cuda:no, cpu]")
    p.add_argument("--batch_size", type=int, help="Batch size")
    p.add_argument("--log_dir", type=int, help="Path to save the model and report")
    p.add_argument("--load_IMU_path", type=str, help="Path to load IMU_Net")
    p.add_argument("--load_Upper_path", type=str, help="Path to load Upper_Net")
    p.add_argument("--load_Lower_path", type=str, help="Path to load Lower_Net")
    p.add_argument("--gt_head_pose", action="store_true", help="head pose from the recording instead of IMU_Net")
    p.add_argument("--data_root", type=str, help="Sample_data directory")
    p.add_argument("--seed", type=int, help="seed torch (net initialisation) and numpy (point-cloud padding) -- the reference does not seed")
    p.add_argument("--imu_precision", type=str, choices=["fp32", "split3", "bf16"],
                   help="eval-mode IMU_Net forwards (stages 2/3, --infer): fp32 (default), split3 (fp32-accurate piece products on the "
                        "bf16 matrix pipe, inside the parity bars: DESIGN.md 7c) or bf16 product operands with fp32 accumulation "
                        "(DESIGN.md 7a, outside them)")
    p.add_argument("--imu_train_precision", type=str, choices=["fp32", "split3"],
                   help="stage-1 IMU_Net training: fp32 (default) or split3 (rnn_fast's projection / input-gradient / weight-gradient "
                        "products as piece products: DESIGN.md 7c)")
    p.add_argument("--resume", type=str, help="continue --train from a checkpoint written by this framework (the model .pth "
                                               "or its .train_state.pth: weights, Adam state, epoch, RNGs)")
    return p


def apply_overrides(args):
    both = (Config, ConfigDemo)
    if args.colab:
        ConfigDemo.colab = True
    if args.epochs is not None:
        Config.epochs = args.epochs
    if args.lr is not None:
        Config.lr = args.lr
    if args.batch_size is not None:
        Config.batch_size = args.batch_size
    if args.log_dir is not None:
        Config.Idx = args.log_dir
    for name, attr in (("device", "device"), ("load_IMU_path", "model_IMU_path"), ("load_Upper_path", "model_upper_path"),
                       ("load_Lower_path", "model_lower_path"), ("data_root", "data_root")):
        v = getattr(args, name)
        if v is not None:
            for c in both:
                setattr(c, attr, v)
    if args.gt_head_pose:
        for c in both:
            c.gt_head_pose = True
    Config.resume_path = args.resume
    if args.imu_precision is not None:
        os.environ["MMEGO_IMU_PRECISION"] = args.imu_precision      # read by IMUNet.__init__
    if args.imu_train_precision is not None:
        os.environ["MMEGO_IMU_TRAIN_PRECISION"] = args.imu_train_precision


def main(argv=None):
    args = build_parser().parse_args(argv)
    apply_overrides(args)
    if args.seed is not None:
        import numpy as np
        torch.manual_seed(args.seed)
        np.random.seed(args.seed)          # PosePC pads / subsamples the point clouds with numpy's global generator
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and args.train:
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        Config.device = "cuda:%d" % local
        Config.data_parallel = True
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
    if args.train:
        if args.network == "IMU_Net":
            from Processor.Train.Train_IMU import MMEgo
            MMEgo().train_imu()
        if args.network == "Upper_Net":
            from Processor.Train.Train_Upper import MMEgo
            MMEgo().train_upper()
        if args.network == "Lower_Net":
            from Processor.Train.Train_Lower import MMEgo
            MMEgo().train_lower()
    elif args.infer:
        from Processor.Test.Demo_test import MMEgo
        processor = MMEgo()
        if args.vis:
            processor.eval_all_skeleton()
        else:
            processor.eval_model()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
