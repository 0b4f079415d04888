"""Stage trainers and the `--infer` evaluator on the HIP path.

Counterparts of the reference's Processor/Train/Train_Upper.py:20-251, Train_Lower.py:23-332,
Train_IMU.py:38-185 and Processor/Test/Demo_test.py:23-184: same class shape (``MMEgo`` with
train_upper / train_lower / train_once / eval_model), same directory side effects
(Processor/Train/{report,model,lossAndacc}/<Idx>/, log-loss.txt, log-eval.txt, epoch*_batch*frame*lr*.pth),
same stdout lines for `--infer`.  Plotting (matplotlib/seaborn) is out of scope.

Data parallel (not in the reference): when launched under torch.distributed.run each rank trains on its
own shard of every epoch's shuffled order; gradients are summed with one RCCL all-reduce per step.
"""
import os

import numpy as np
import torch

from . import hip, ops
from .config import Config, ConfigDemo
from .data import DeviceArrays, PosePC, batch_indices, batches
from .nets import IMUNet, LowerNet, UpperNet
from .train_step import PipelinedStages, StageStep, broadcast_flag, empty_step, shard_of, sync_replicas
from .utils import EarlyStopping

_HERE = os.path.dirname(os.path.abspath(__file__))
# where report/, model/, lossAndacc/ go: the reference's Processor/Train (Train_Upper.py:22-50) unless MMEGO_TRAIN_DIR says otherwise
_TRAIN_DIR = os.environ.get("MMEGO_TRAIN_DIR") or os.path.join(os.path.dirname(_HERE), "Processor", "Train")


def _dev_tensor(a, device):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(device)


class _Base:
    """Shared set-up: directories, frozen IMU_Net (or ground-truth head pose), datasets."""

    def __init__(self, cfg, make_dirs=True):
        self.cfg = cfg
        self.device = torch.device(cfg.device)
        if self.device.type != "cuda":
            raise RuntimeError("the mmego_amd processors run on the MI355X HIP path only (device=%s)" % cfg.device)
        hip.lib()
        self.frame_no, self.batchsize = cfg.frame_no, getattr(cfg, "batch_size", 20)
        self.Idx = cfg.Idx
        self.world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
        self.rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
        self.pg = torch.distributed.group.WORLD if self.world > 1 else None
        if make_dirs and self.rank == 0:
            for sub in ("report", "model", "lossAndacc"):
                path = os.path.join(_TRAIN_DIR, sub, str(self.Idx))
                os.makedirs(path, exist_ok=True)
                print("%s saved in %s" % ({"report": "report", "model": "model", "lossAndacc": "Loss and accuracy"}[sub], path))

    def _dp_start(self, model):
        """Data parallel: every replica starts from rank 0's weights and buffers (the nets are built from each rank's own
        torch RNG unless --seed is given), and the dropout streams of the ranks are decorrelated (SURVEY 8-e ii)."""
        if self.world > 1:
            sync_replicas(model, self.pg)
            model.seed_counter().bitwise_xor_((self.rank * 0x9E3779B97F4A7C15) & 0x7FFFFFFFFFFFFFFF)

    def _log_mode(self):
        return "a" if getattr(self.cfg, "resume_path", None) else "w"     # --resume continues the logs instead of truncating them

    def _load_imu(self):
        imu = IMUNet(15, 6 + 3, 512, 2, True, 0.1).to(self.device).eval()
        if self.cfg.gt_head_pose:
            print("[mmego_amd] head pose from the recording (R_R0R, ground-truth head joint); IMU_Net not used")
            return None
        if not os.path.exists(self.cfg.model_IMU_path):
            raise FileNotFoundError("IMU_Net checkpoint %s not found (it is absent from the reference snapshot); pass "
                                    "--load_IMU_path or --gt_head_pose" % self.cfg.model_IMU_path)
        imu.load(self.cfg.model_IMU_path)
        return imu

    def head_pose(self, imu_net, imu, R_R0R, target):
        """(R, t) fed to Upper/Lower: frozen IMU_Net output, or the recorded pose when --gt_head_pose."""
        if imu_net is None:
            return R_R0R.contiguous(), target[:, :, 20].contiguous()
        with torch.no_grad():
            return imu_net(imu)

    def save_models(self, epoch, model, optimizer=None, early=None):
        """The reference's checkpoint (state_dict only, same file name: Train_Upper.py:69-74) plus, beside it, a
        `.train_state.pth` with what a bit-exact `--resume` needs and the reference never saved: Adam moments and step
        count, the epoch, the minibatch-order RNG, the dropout counter and the early-stopping state."""
        if self.rank != 0:
            return None
        path = os.path.join(_TRAIN_DIR, "model", str(self.Idx), "epoch{}_batch{}frame{}lr{}.pth".format(
            epoch, self.batchsize, self.frame_no, self.learning_rate))
        torch.save(model.state_dict(), path)
        if optimizer is not None:
            torch.save({"epoch": epoch + 1, "model": model.state_dict(), "optimizer": optimizer.state_dict(),
                        "rng": self._rng.get_state(), "dropout_counter": model.seed_counter().cpu(),
                        "early": None if early is None else (early.counter, early.best_score)},
                       path[:-4] + ".train_state.pth")
        return path

    def load_train_state(self, path, model):
        """Restore a `.train_state.pth` written by save_models (or the one beside a given model `.pth`)."""
        if not path.endswith(".train_state.pth"):
            path = path[:-4] + ".train_state.pth"
        ts = torch.load(path, map_location="cpu", weights_only=False)
        model.load_state_dict(ts["model"])
        model.seed_counter().copy_(ts["dropout_counter"])
        self._rng.set_state(ts["rng"])
        self.start_epoch = int(ts["epoch"])
        self._resume = ts
        print("[mmego_amd] resumed from %s at epoch %d" % (path, self.start_epoch))
        return ts

    def pose_metrics(self, upper_l, lower_l, target):
        """Demo_test per-batch figures on the device -> (all, upper, lower, per_joint[21], angle[20])."""
        F = upper_l.shape[0] * upper_l.shape[1]
        E = torch.empty((F, 43), dtype=torch.float32, device=self.device)
        hip.call("pose_errors", upper_l.contiguous(), lower_l.contiguous(), target.contiguous(), F, E)
        s = torch.empty(43, dtype=torch.float32, device=self.device)
        ops.colsum(E, s)
        m = (s / F).cpu().numpy().astype(np.float64)
        return float(m[:21].mean()), float(m[41]), float(m[42]), m[:21], m[21:41]


class _StageTrainer(_Base):
    stage = None

    def __init__(self):
        super().__init__(Config)
        cfg = self.cfg
        self.num_epochs, self.save_slot, self.learning_rate = cfg.epochs, 50, cfg.lr
        self.model_IMU = self._load_imu()
        self.train_data = PosePC(batch_length=self.frame_no)
        self.test_data = PosePC(train=False, batch_length=self.frame_no)
        rep = os.path.join(_TRAIN_DIR, "report", str(self.Idx))
        self.lossfile = open(os.path.join(rep, "log-loss.txt"), self._log_mode()) if self.rank == 0 else None
        self.evalfile = open(os.path.join(rep, "log-eval.txt"), self._log_mode()) if self.rank == 0 else None
        self._steps = {}
        self._rng = np.random.RandomState(1234)
        self.start_epoch, self._resume = 0, None
        self._train_dev = None
        self._log = None
        self._opt_shared = None

    def _optimizer(self):
        return self._opt_shared

    def _step_for(self, B):
        """One StageStep (static buffers, optional HIP graph) per minibatch size.  With a frozen IMU_Net in the loop its forward
        runs one minibatch ahead of the trainable body (train_step.PipelinedStages: same results bit for bit, the compute-bound
        IMU_Net forward of minibatch i+1 overlaps the latency-bound body of minibatch i); MMEGO_PIPELINE_IMU=0 turns that off."""
        st = self._steps.get(B)
        if st is None:
            pg = self.pg
            pipelined = self.model_IMU is not None and os.environ.get("MMEGO_PIPELINE_IMU", "1") != "0"
            st = StageStep(self.stage, self.model, None if pipelined else self.model_IMU,
                           upper_frozen=getattr(self, "Upper_net", None), lr=self.learning_rate, process_group=pg,
                           use_graph=not pipelined)
            if self._opt_shared is not None:
                st.opt = self._opt_shared                               # one optimiser state for all batch sizes
            else:
                self._opt_shared = st.opt
                if self._resume is not None:
                    st.opt.load_state_dict(self._resume["optimizer"])
            st.engine = None
            if pipelined:
                st.imu_next = torch.empty((B, self.frame_no) + tuple(self._train_dev.shape["imu"][1:]), dtype=torch.float32,
                                          device=self.device)
                st.engine = PipelinedStages([st], [self.model_IMU], st.imu_next, use_graph=True)
            self._steps[B] = st
        return st

    def train_once(self):
        self.model.train()
        nsel = self.model_out_joints
        if self._log is None:
            self._log = torch.zeros((64, 2), dtype=torch.float32, device=self.device)
        nlog, scales = 0, []
        if self._train_dev is None:                                     # the training set lives in HBM (51 MB for Sample_data)
            self._train_dev = DeviceArrays(self.train_data, self.device)
        todo = [idx[shard_of(self.rank, self.world)]                    # this rank's shard of every global minibatch
                for idx in batch_indices(len(self.train_data), self.batchsize * self.world, True, self._rng)]
        primed = None                                                   # number of the minibatch whose head poses are ready
        for i, idx in enumerate(todo):
            B = len(idx)
            if B == 0:              # short last global minibatch, nothing for this rank: zero gradient, same collective + update
                if self._opt_shared is None:
                    from .params import FusedAdam
                    self._opt_shared = FusedAdam(self.model.flat(), lr=self.learning_rate)
                    if self._resume is not None:
                        self._opt_shared.load_state_dict(self._resume["optimizer"])
                empty_step(self.model, self._opt_shared, self.pg)
                primed = None
                continue
            st = self._step_for(B)
            b = self._train_dev.gather(idx)                             # on-device gather into per-batch-size static buffers
            tgt = b["target"]
            if st.static is None or st.static["x_src"].data_ptr() != b["data"].data_ptr():
                st.bind(b["data"], b["imu"], b["skl"], tgt, R_gt=b["R_R0R"])
            if st.engine is None:
                st.step()
            else:
                if primed != i:                                         # first minibatch, or the size changed: no forward ran ahead
                    st.imu_next.copy_(b["imu"])
                    st.engine.prime()
                if i + 1 < len(todo) and len(todo[i + 1]) == B:         # IMU samples of the next minibatch for the forward that
                    self._train_dev.gather_field_into("imu", todo[i + 1], st.imu_next)      # runs beside this body
                    primed = i + 1
                else:
                    primed = None
                st.engine.step()
            # per-minibatch log (L1 sum, mean joint distance): kept on the device, read once per epoch (the reference
            # calls loss.item() every step: Train_Upper.py:183-186)
            if nlog == self._log.shape[0]:
                self._log = torch.cat((self._log, torch.zeros_like(self._log)))
            ops.copy2d(st.loss2.view(1, 2), self._log[nlog:nlog + 1])
            scales.append(1.0 / (B * self.frame_no * nsel))
            nlog += 1
        log = self._log[:nlog].cpu().numpy()
        losses = [float(v) for v in log[:, 0]]
        accs = [float(v * sc) for v, sc in zip(log[:, 1], scales)]
        assert nsel in (15, 8)
        return accs, losses

    def _train_loop(self, extra_print):
        early = EarlyStopping(patience=30)
        if getattr(self.cfg, "resume_path", None):
            ts = self.load_train_state(self.cfg.resume_path, self.model)
            if ts["early"] is not None:
                early.counter, early.best_score = ts["early"]
        out = None
        for epoch in range(self.start_epoch, self.num_epochs):
            print("epoch: {}".format(epoch + 1))
            self.train_once()
            sync_replicas(self.model, self.pg, params=False)       # BatchNorm running statistics: rank 0's, on every rank
            out = self.eval_model()
            eval_loss, eval_loss_l, eval_accu, second, accu_ll, angle_ll = out
            if self.rank == 0:
                self.lossfile.write("%d %f\n" % (epoch + 1, eval_loss))
                self.lossfile.write(str(eval_loss_l) + "\n")
                self.lossfile.flush()
                extra_print(epoch, out)
            # every rank evaluates its own copy of the test split (the reference's unseeded padding makes them differ), so
            # the ranks could disagree on when to stop and leave each other hanging in the next all-reduce: rank 0 decides
            stop = broadcast_flag(early(eval_loss), self.device, self.pg)
            # (the reference saves before the evaluation pass; saving after it keeps the RNG / early-stopping state in the
            #  checkpoint consistent with "epoch finished", which is what --resume continues from)
            if (epoch + 1) % self.save_slot == 0 or epoch + 1 == self.num_epochs or stop:
                self.save_models(epoch, self.model, self._optimizer(), early)
            if stop:
                print("Early stopping")
                break
        return out


class UpperTrainer(_StageTrainer):
    """`python main.py --train --network Upper_Net` (reference Train_Upper.MMEgo)."""
    stage = "upper"
    model_out_joints = 15

    def __init__(self):
        super().__init__()
        self.model = UpperNet().to(self.device)
        if self.cfg.Upper_pretrained:
            self.model.load(self.cfg.model_upper_path)
        self._dp_start(self.model)

    def train_upper(self):
        def report(epoch, out):
            eval_loss, eval_loss_l, eval_accu, dis_l, accu_ll, angle_ll = out
            self.evalfile.write("%d %f %f\n" % (epoch + 1, eval_accu, sum(angle_ll) / len(angle_ll)))
            self.evalfile.write(str(accu_ll) + "\n" + str(angle_ll) + "\n" + str(accu_ll[self.cfg.hand_joint_map]) + "\n")
            self.evalfile.flush()
            print("\n wrist elbow(l, r): {}".format(accu_ll[self.cfg.hand_joint_map]))
            print("Average Joint Localization Error: {}".format(eval_accu))
            print("Average Joint Rotation Error: {}".format(sum(angle_ll) / len(angle_ll)))
            print(accu_ll)
            print("Eval_loss: {} Eval_loss_l (l, g): {} \n Angle_loss: {} \n".format(eval_loss, eval_loss_l, angle_ll))
        return self._train_loop(report)

    def eval_model(self):
        self.model.eval()
        dev = self.device
        umap = self.cfg.upper_joint_map
        loss_l, accu_l, per_joint, angles, dis = [], [], [], [], []
        bones = [(umap.index(p), umap.index(c)) for p, c in self.cfg.skeleton_upper_body.tolist()]
        with torch.no_grad():
            for data, target, skl, imu, _, _, R_R0R, _ in batches(self.test_data, self.batchsize, True, self._rng):
                B, T = data.shape[0], data.shape[1]
                x, tgt = _dev_tensor(data, dev), _dev_tensor(target, dev)
                R, t = self.head_pose(self.model_IMU, _dev_tensor(imu, dev), _dev_tensor(R_R0R, dev), tgt)
                h0 = torch.zeros((6, B, 64), device=dev)
                up = self.model(x, h0, h0.clone(), _dev_tensor(skl, dev), R, t)[0]
                tu = tgt[:, :, umap, :]
                loss_l.append((up - tu).abs().sum().item() / B / T)
                d = torch.sqrt(torch.sum(torch.square(up - tu), dim=-1))
                accu_l.append(d.mean().item())
                per_joint.append(d.mean(0).mean(0).cpu().numpy())
                dis.append((up - tu).abs().mean().item())
                pv = torch.stack([up[:, :, c] - up[:, :, p] for p, c in bones], 2)
                tv = torch.stack([tu[:, :, c] - tu[:, :, p] for p, c in bones], 2)
                cs = torch.nn.functional.cosine_similarity(pv, tv, dim=-1)
                angles.append((torch.acos(cs.clamp(-1.0, 1.0)) / 3.14159265358 * 180.0).abs().mean(0).mean(0).cpu().numpy())
        eval_loss = float(np.mean(loss_l))
        return (eval_loss, np.asarray([eval_loss / self.cfg.joint_num_upper]), float(np.mean(accu_l)), float(np.mean(dis)),
                np.mean(per_joint, axis=0), np.mean(angles, axis=0))


class LowerTrainer(_StageTrainer):
    """`python main.py --train --network Lower_Net` (reference Train_Lower.MMEgo)."""
    stage = "lower"
    model_out_joints = 8

    def __init__(self):
        super().__init__()
        self.model = LowerNet(hidden_dim=64).to(self.device)
        if self.cfg.Lower_pretrained:
            self.model.load(self.cfg.model_lower_path)
        self.Upper_net = UpperNet().to(self.device).eval()
        self.Upper_net.load(self.cfg.model_upper_path)
        self._dp_start(self.model)

    def train_lower(self):
        def report(epoch, out):
            eval_loss, eval_loss_l, eval_accu, accu_lower, accu_ll, angle_ll = out
            self.evalfile.write("%d %f %f %f\n" % (epoch + 1, eval_accu, accu_lower, sum(angle_ll) / len(angle_ll)))
            self.evalfile.write(str(accu_ll) + "\n" + str(angle_ll) + "\n" + str(accu_ll[self.cfg.hand_joint_map]) + "\n")
            self.evalfile.flush()
            print("\n wrist elbow(l, r): {}".format(accu_ll[self.cfg.hand_joint_map]))
            print("Average Joint Localization Error: {}".format(eval_accu))
            print("Average LowerBody Joint Localization Error: {}".format(accu_lower))
            print("Average Joint Rotation Error: {}".format(sum(angle_ll) / len(angle_ll)))
            print(accu_ll)
            print("Eval_loss: {} Eval_loss_l (l, g): {} \n Angle_loss: {}".format(eval_loss, eval_loss_l, angle_ll))
        return self._train_loop(report)

    def eval_model(self):
        self.model.eval()
        return evaluate_full(self, self.model_IMU, self.Upper_net, self.model, batches(self.test_data, self.batchsize, True, self._rng))[0]


def evaluate_full(base, imu_net, upper_net, lower_net, batch_iter):
    """IMU -> Upper -> Lower over an iterator of minibatches; returns (reference-style tuple, summary dict)."""
    dev = base.device
    cfg = base.cfg
    loss_l, accu_l, accu_up, accu_lo, per_joint, angles = [], [], [], [], [], []
    lmap = cfg.lower_joint_map
    with torch.no_grad():
        for batch in batch_iter:
            data, target, skl, imu, R_R0R = batch[0], batch[1], batch[2], batch[3], batch[6]
            B, T = data.shape[0], data.shape[1]
            x, tgt = _dev_tensor(data, dev), _dev_tensor(target, dev)
            R, t = base.head_pose(imu_net, _dev_tensor(imu, dev), _dev_tensor(R_R0R, dev), tgt)
            h0 = torch.zeros((6, B, 64), device=dev)
            body = _dev_tensor(skl, dev)
            up = upper_net(x, h0, h0.clone(), body, R, t)[0]
            lo, _ = lower_net(up.clone(), x, h0, h0, h0, h0, body, R, t)        # x already transformed once (Q1)
            loss_l.append((lo - tgt[:, :, lmap, :]).abs().sum().item() / B / T)
            a, u, l, pj, ang = base.pose_metrics(up, lo, tgt)
            accu_l.append(a); accu_up.append(u); accu_lo.append(l); per_joint.append(pj); angles.append(ang)
    eval_loss = float(np.mean(loss_l))
    accu_ll, angle_ll = np.mean(per_joint, axis=0), np.mean(angles, axis=0)
    summary = dict(all_cm=float(np.mean(accu_l)) * 100, upper_cm=float(np.mean(accu_up)) * 100, lower_cm=float(np.mean(accu_lo)) * 100,
                   rot_deg=float(sum(angle_ll) / len(angle_ll)), per_joint_cm=accu_ll * 100)
    return (eval_loss, np.asarray([eval_loss / cfg.joint_num_lower]), float(np.mean(accu_l)), float(np.mean(accu_lo)), accu_ll, angle_ll), summary


class Evaluator(_Base):
    """`python main.py --infer` (reference Processor/Test/Demo_test.MMEgo.eval_model)."""

    def __init__(self):
        super().__init__(ConfigDemo, make_dirs=False)
        cfg = self.cfg
        self.model = LowerNet(hidden_dim=64).to(self.device).eval()
        self.model_IMU = self._load_imu()
        self.Upper_net = UpperNet().to(self.device).eval()
        self.Upper_net.load(cfg.model_upper_path)
        self.vis_data = PosePC(train=False, vis=True, batch_length=self.frame_no)

    def eval_model(self):
        self.model.load(self.cfg.model_lower_path)
        self.model.eval()
        out, s = evaluate_full(self, self.model_IMU, self.Upper_net, self.model, batches(self.vis_data, 1, False))
        print("Average Joint Localization Error(cm): {}".format(s["all_cm"]))
        print("Average UpperBody Joint Localization Error(cm): {}".format(s["upper_cm"]))
        print("Average LowerBody Joint Localization Error(cm): {}".format(s["lower_cm"]))
        print("Average Joint Rotation Error(°): {}".format(s["rot_deg"]))
        print("Per Joint Localization Error(cm): {}".format(s["per_joint_cm"]))
        return out

    def eval_all_skeleton(self):
        raise NotImplementedError("--vis (matplotlib animation) is outside the hot path and not provided")


class ImuTrainer(_Base):
    """`python main.py --train --network IMU_Net` (reference Train_IMU.MMEgo): stage 1, geodesic + 100 x position loss,
    Adam with coupled weight decay 1e-3."""

    def __init__(self):
        super().__init__(Config)
        cfg = self.cfg
        self.num_epochs, self.save_slot, self.learning_rate = cfg.epochs, 50, cfg.lr
        self.model_IMU = IMUNet(15, 6 + 3, 512, 2, True, 0).to(self.device)
        if cfg.IMU_pretrained:
            self.model_IMU.load(cfg.model_IMU_path)
        from .params import FusedAdam
        self._dp_start(self.model_IMU)
        self.optimizer_IMU = FusedAdam(self.model_IMU.flat(), lr=self.learning_rate, weight_decay=0.001)
        self.train_data = PosePC(batch_length=self.frame_no)
        self.test_data = PosePC(train=False, batch_length=self.frame_no)
        rep = os.path.join(_TRAIN_DIR, "report", str(self.Idx))
        self.lossfile = open(os.path.join(rep, "log-loss.txt"), self._log_mode()) if self.rank == 0 else None
        self._rng = np.random.RandomState(1234)
        self._loss = torch.zeros(1, device=self.device)
        self.start_epoch, self._resume = 0, None
        self._train_dev = None
        self._steps = {}

    def _loss_and_grads(self, R, t, R_gt, head, want_grad):
        F = R.shape[0] * R.shape[1]
        dR = torch.empty_like(R) if want_grad else None
        dt = torch.empty_like(t) if want_grad else None
        hip.call("imu_loss", R.contiguous(), t.contiguous(), R_gt.contiguous(), head.contiguous(), F, 1.0, self._loss, dR, dt)
        return dR, dt

    def train_imu_once(self):
        from .train_step import ImuStep
        self.model_IMU.train()
        losses = []
        pg = self.pg
        if self._train_dev is None:
            self._train_dev = DeviceArrays(self.train_data, self.device)
        for idx in batch_indices(len(self.train_data), self.batchsize * self.world, True, self._rng):
            idx = idx[shard_of(self.rank, self.world)]
            if len(idx) == 0:       # nothing for this rank in a short last global minibatch: zero gradient, same update
                empty_step(self.model_IMU, self.optimizer_IMU, pg)
                continue
            b = self._train_dev.gather(idx)
            B, T = b["imu"].shape[0], b["imu"].shape[1]
            st = self._steps.get(B)
            if st is None:                                              # one graph per minibatch size, one optimiser
                st = ImuStep(self.model_IMU, lr=self.learning_rate, weight_decay=0.001, process_group=pg, use_graph=True)
                st.opt = self.optimizer_IMU
                self._steps[B] = st
            if st.static is None or st.static["imu"].data_ptr() != b["imu"].data_ptr():
                st.bind(b["imu"], b["R_R0R"], b["target"])
            st.step()
            losses.append(st.loss.item() / B / T)
        return float(np.mean(losses))

    def eval_imu(self):
        self.model_IMU.eval()
        tot, parts = [], []
        with torch.no_grad():
            for data, target, skl, imu, _, _, R_R0R, _ in batches(self.test_data, self.batchsize, True, self._rng):
                dev = self.device
                B, T = imu.shape[0], imu.shape[1]
                tgt, Rg = _dev_tensor(target, dev), _dev_tensor(R_R0R, dev)
                R, t = self.model_IMU(_dev_tensor(imu, dev))
                self._loss_and_grads(R, t, Rg, tgt[:, :, 20], False)
                loss = self._loss.item()
                pos = torch.sqrt(((t - tgt[:, :, 20]) ** 2).sum(-1)).sum().item()
                tot.append(loss / B / T)
                parts.append([(loss - 100 * pos) / B / T, pos / B / T])
        return float(np.mean(tot)), np.mean(parts, axis=0)

    def train_imu(self):
        early = EarlyStopping(patience=30)
        if getattr(self.cfg, "resume_path", None):
            ts = self.load_train_state(self.cfg.resume_path, self.model_IMU)
            self.optimizer_IMU.load_state_dict(ts["optimizer"])
            if ts["early"] is not None:
                early.counter, early.best_score = ts["early"]
        for epoch in range(self.start_epoch, self.num_epochs):
            print("epoch: {}".format(epoch + 1))
            train_loss = self.train_imu_once()
            eval_loss, eval_loss_l = self.eval_imu()
            if self.rank == 0:
                self.lossfile.write("%d %f\n" % (epoch + 1, eval_loss))
                self.lossfile.write(str(eval_loss_l) + "\n")
                self.lossfile.flush()
            print("Train_loss: {}".format(train_loss))
            print("Eval_loss: {}  Eval_loss_l (angle, H_pos): {}".format(eval_loss, eval_loss_l))
            stop = broadcast_flag(early(eval_loss), self.device, self.pg)       # rank 0 decides for everybody
            if (epoch + 1) % self.save_slot == 0 or epoch + 1 == self.num_epochs or stop:
                self.save_models(epoch, self.model_IMU, self.optimizer_IMU, early)
            if stop:
                print("Early stopping")
                break
