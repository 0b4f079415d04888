"""GPU: the reference's command line end to end on a tiny synthetic Sample_data tree (same .mat keys and layout):
stage-2, stage-3 and stage-1 training for one epoch each, then --infer with the checkpoints just written."""
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.io as scio

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _make_dataset(root, rng):
    for a in (1, 2):
        for s in (1, 2, 3):
            d = os.path.join(root, "%02d" % a, "s%d" % s)
            os.makedirs(d)
            skel = rng.normal(0, 0.4, (32, 3)) + np.array([0.8, 0.0, 0.2])
            for f in range(23):
                n = int(rng.integers(20, 150))
                pc = np.concatenate([rng.normal([0.8, 0.0, 0.2], 0.4, (n, 3)), rng.uniform(10, 46, (n, 1)), rng.normal(0, 0.4, (n, 1))], 1)
                q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
                imu = np.concatenate([np.tile(q.reshape(1, 9), (20, 1)), rng.normal(size=(20, 6))], 1)
                scio.savemat(os.path.join(d, "frame_%d.mat" % f), {
                    "pc_xyziv_ti2": pc.astype(np.float32), "pc_xyz_key_2": skel + rng.normal(0, 0.01, (32, 3)),
                    "imu_save_l": imu, "R_btc": q, "orientation_imu_img": np.eye(3), "t_R0R": rng.normal(size=(1, 3)),
                    "abcd_ground_2": np.array([[0.0, 0.0, -1.0, 1.0]]), "foot_contact": np.array([[1, 0]])})


def _run(args, env):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-3000:]
    return r.stdout


def test_main_train_and_infer_on_synthetic_tree(tmp_path):
    data = str(tmp_path / "Sample_data")
    _make_dataset(data, np.random.default_rng(0))
    out_dir = str(tmp_path / "train_out")
    env = dict(os.environ, PYTHONPATH=ROOT, MMEGO_TRAIN_DIR=out_dir)
    model_dir = os.path.join(out_dir, "model")
    common = ["--data_root", data, "--epochs", "1", "--batch_size", "4", "--device", "cuda:0"]
    out = _run(["--train", "--network", "Upper_Net", "--gt_head_pose", "--log_dir", "9101"] + common, env)
    assert "epoch: 1" in out and "Average Joint Localization Error" in out
    # the trainers save on (epoch+1) % 50 == 0 or early stop; write checkpoints for the next stages from fresh nets
    import torch
    from mmego_amd import nets
    ck = tmp_path / "ck"
    ck.mkdir()
    torch.manual_seed(0)
    torch.save(nets.UpperNet().state_dict(), ck / "upper.pth")
    torch.save(nets.LowerNet(64).state_dict(), ck / "lower.pth")
    out = _run(["--train", "--network", "Lower_Net", "--gt_head_pose", "--log_dir", "9102", "--load_Upper_path", str(ck / "upper.pth")] + common, env)
    assert "Average LowerBody Joint Localization Error" in out
    out = _run(["--train", "--network", "IMU_Net", "--log_dir", "9103"] + common, env)
    assert "Train_loss:" in out and "Eval_loss:" in out
    out = _run(["--infer", "--gt_head_pose", "--data_root", data, "--device", "cuda:0", "--load_Upper_path", str(ck / "upper.pth"),
                "--load_Lower_path", str(ck / "lower.pth")], env)
    for line in ("Average Joint Localization Error(cm):", "Average UpperBody Joint Localization Error(cm):",
                 "Average LowerBody Joint Localization Error(cm):", "Average Joint Rotation Error", "Per Joint Localization Error(cm):"):
        assert line in out
    assert os.path.exists(os.path.join(out_dir, "report", "9101", "log-loss.txt"))
    assert os.path.isdir(model_dir)
    # head pose from IMU_Net instead of the recording, fp32 and with --imu_precision bf16 (opt-in mode): the
    # bf16 run moves the head rotation by ~2e-3, i.e. the average joint error by well under a centimetre
    torch.save(nets.IMUNet(15, 9, 512, 2, True, 0.1).state_dict(), ck / "imu.pth")
    import re
    errs = {}
    for prec in ("fp32", "bf16"):
        out = _run(["--infer", "--data_root", data, "--device", "cuda:0", "--load_Upper_path", str(ck / "upper.pth"),
                    "--load_Lower_path", str(ck / "lower.pth"), "--load_IMU_path", str(ck / "imu.pth"), "--imu_precision", prec], env)
        errs[prec] = float(re.search(r"Average Joint Localization Error\(cm\):\s*([0-9.eE+-]+)", out).group(1))
    assert errs["fp32"] != errs["bf16"] and abs(errs["fp32"] - errs["bf16"]) < 0.5, errs


def test_resume_continues_bit_exactly(tmp_path):
    """`--resume`: 1 epoch + resume for the 2nd == 2 epochs in one go, bit for bit (weights, BN buffers):
    the checkpoint carries Adam moments and step count, the minibatch-order RNG, the dropout counter, early-stopping state."""
    import glob
    import torch
    data = str(tmp_path / "Sample_data")
    _make_dataset(data, np.random.default_rng(1))
    out_dir = str(tmp_path / "train_out")
    env = dict(os.environ, PYTHONPATH=ROOT, MMEGO_TRAIN_DIR=out_dir)
    mdir = os.path.join(out_dir, "model")
    base = ["--train", "--network", "Upper_Net", "--gt_head_pose", "--data_root", data, "--batch_size", "4", "--device", "cuda:0",
            "--seed", "5"]
    _run(base + ["--epochs", "2", "--log_dir", "9111"], env)
    _run(base + ["--epochs", "1", "--log_dir", "9112"], env)
    first = glob.glob(os.path.join(mdir, "9112", "epoch0_*lr*.pth"))
    first = [f for f in first if not f.endswith(".train_state.pth")]
    assert len(first) == 1 and os.path.exists(first[0][:-4] + ".train_state.pth")
    out = _run(base + ["--epochs", "2", "--log_dir", "9113", "--resume", first[0]], env)
    assert "resumed from" in out and "epoch: 2" in out and "epoch: 1\n" not in out
    a = [f for f in glob.glob(os.path.join(mdir, "9111", "epoch1_*.pth")) if not f.endswith(".train_state.pth")]
    b = [f for f in glob.glob(os.path.join(mdir, "9113", "epoch1_*.pth")) if not f.endswith(".train_state.pth")]
    assert len(a) == 1 and len(b) == 1
    sa, sb = torch.load(a[0], map_location="cpu"), torch.load(b[0], map_location="cpu")
    assert sa.keys() == sb.keys()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    ta = torch.load(a[0][:-4] + ".train_state.pth", map_location="cpu", weights_only=False)
    tb = torch.load(b[0][:-4] + ".train_state.pth", map_location="cpu", weights_only=False)
    assert torch.equal(ta["optimizer"]["m"], tb["optimizer"]["m"]) and torch.equal(ta["optimizer"]["state"], tb["optimizer"]["state"])


def test_training_with_imu_net_pipelined_equals_unpipelined(tmp_path):
    """`--train --network Upper_Net` with the head pose from a frozen IMU_Net: by default the trainer runs the IMU_Net forward one
    minibatch ahead of the trainable body (train_step.PipelinedStages); MMEGO_PIPELINE_IMU=0 runs it inside the body.  Two epochs
    with minibatches of 3, 3 and 2 windows (a size change mid-epoch, where nothing runs ahead): identical checkpoints."""
    import glob
    import torch
    from mmego_amd import nets
    data = str(tmp_path / "Sample_data")
    _make_dataset(data, np.random.default_rng(2))
    torch.manual_seed(1)
    imu_ck = str(tmp_path / "imu.pth")
    torch.save(nets.IMUNet(15, 9, 512, 2, True, 0.1).state_dict(), imu_ck)
    out_dir = str(tmp_path / "train_out")
    model_dir = os.path.join(out_dir, "model")
    sd = {}
    for tag, idx, extra in (("pipelined", "9121", {}), ("plain", "9122", {"MMEGO_PIPELINE_IMU": "0"})):
        env = dict(os.environ, PYTHONPATH=ROOT, MMEGO_TRAIN_DIR=out_dir, **extra)
        out = _run(["--train", "--network", "Upper_Net", "--load_IMU_path", imu_ck, "--data_root", data, "--epochs", "2",
                    "--batch_size", "3", "--device", "cuda:0", "--seed", "3", "--log_dir", idx], env)
        assert "epoch: 2" in out
        files = sorted(glob.glob(os.path.join(model_dir, idx, "epoch*_batch3frame*.pth")))      # (saved once, after the last epoch)
        files = [f for f in files if not f.endswith(".train_state.pth")]
        assert files, os.listdir(os.path.join(model_dir, idx))
        sd[tag] = torch.load(files[-1], map_location="cpu")
    assert sd["pipelined"].keys() == sd["plain"].keys()
    for k in sd["plain"]:
        assert torch.equal(sd["pipelined"][k], sd["plain"][k]), k


def test_training_with_the_frozen_imu_net_in_split3_mode(tmp_path):
    """`main.py --train --network Upper_Net --imu_precision split3`: the frozen IMU_Net's forwards on fp32-accurate piece products
    (DESIGN.md 7c), pipelined and unpipelined: identical checkpoints, and within fp32 rounding's reach of the default run's after two
    epochs (the head poses differ in the last bits, the trained weights follow)."""
    import glob
    import torch
    from mmego_amd import nets
    data = str(tmp_path / "Sample_data")
    _make_dataset(data, np.random.default_rng(2))
    torch.manual_seed(1)
    imu_ck = str(tmp_path / "imu.pth")
    torch.save(nets.IMUNet(15, 9, 512, 2, True, 0.1).state_dict(), imu_ck)
    out_dir = str(tmp_path / "train_out")
    model_dir = os.path.join(out_dir, "model")
    sd = {}
    for tag, idx, args, extra in (("split3", "9131", ["--imu_precision", "split3"], {}),
                                  ("split3_plain", "9132", ["--imu_precision", "split3"], {"MMEGO_PIPELINE_IMU": "0"}),
                                  ("fp32", "9133", [], {})):
        env = dict(os.environ, PYTHONPATH=ROOT, MMEGO_TRAIN_DIR=out_dir, **extra)
        env.pop("MMEGO_IMU_PRECISION", None)
        out = _run(["--train", "--network", "Upper_Net", "--load_IMU_path", imu_ck, "--data_root", data, "--epochs", "2",
                    "--batch_size", "3", "--device", "cuda:0", "--seed", "3", "--log_dir", idx] + args, env)
        assert "epoch: 2" in out
        files = [f for f in sorted(glob.glob(os.path.join(model_dir, idx, "epoch*_batch3frame*.pth"))) if not f.endswith(".train_state.pth")]
        assert files, os.listdir(os.path.join(model_dir, idx))
        sd[tag] = torch.load(files[-1], map_location="cpu")
    for k in sd["fp32"]:
        assert torch.equal(sd["split3"][k], sd["split3_plain"][k]), k
        if sd["fp32"][k].is_floating_point():
            assert torch.allclose(sd["split3"][k], sd["fp32"][k], rtol=1e-3, atol=2e-4), (k, float((sd["split3"][k] - sd["fp32"][k]).abs().max()))
