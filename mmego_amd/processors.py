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

from . import blocks, hip, ops
from .config import Config, ConfigDemo
from .data import DeviceArrays, PosePC, batch_indices
from .nets import IMUNet, LowerNet, UpperNet
from .train_step import PipelinedStages, StageStep, broadcast_flag, empty_step, shard_of, sync_replicas
from .utils import EarlyStopping

_HERE = os.path.dirname(os.path.abspath(__file__))
# where report/, model/, lossAndacc/ go: the reference's Processor/Train (Train_Upper.py:22-50) unless MMEGO_TRAIN_DIR says otherwise
_TRAIN_DIR = os.environ.get("MMEGO_TRAIN_DIR") or os.path.join(os.path.dirname(_HERE), "Processor", "Train")


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
            # a persistent rnn_slow launch of the frozen IMU_Net whose workgroups were not co-resident reports it only through a
            # sticky word: read once per epoch (train_once has just synchronised for its log) -- never train on silently
            blocks.seq_xcd_raise()
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
        """Reference Processor/Train/Train_Upper.py:189-251, on the device end to end: the test split lives in HBM (DeviceArrays,
        uploaded once), minibatches are gathered there, the per-frame figures come from one kernel (mmego_pose_errors_upper), their
        per-minibatch sums land in a device-side log, and the host reads that log ONCE per epoch.  Returns the reference's tuple
        (eval_loss, eval_loss_l, eval_accu, dis_l, accu_ll[15], angle_ll[14]): means over minibatches of per-minibatch means."""
        self.model.eval()
        m = _epoch_eval(self, self.test_data, self.batchsize, True, self._rng, self.model_IMU, self.model, None)
        eval_loss = float(np.mean(m[:, 29]))                                 # L1(sum) / B / T per minibatch
        return (eval_loss, np.asarray([eval_loss / self.cfg.joint_num_upper]), float(np.mean(m[:, :15].mean(axis=1))),
                float(np.mean(m[:, 29] / 45.0)), m[:, :15].mean(axis=0), m[:, 15:29].mean(axis=0))


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
        return evaluate_full(self, self.model_IMU, self.Upper_net, self.model, self.test_data, self.batchsize, True, self._rng)[0]


def _epoch_eval(base, dataset, batch_size, shuffle, rng, imu_net, upper_net, lower_net):
    """One evaluation pass with ONE host synchronisation: -> float64 array [n_minibatches, W] of per-frame MEANS per minibatch.
    lower_net None: W = 30, the columns of mmego_pose_errors_upper; else W = 45: the 43 columns of mmego_pose_errors, then the
    Lower stage's L1(sum) loss per frame and its mean joint distance.  The split is uploaded once per dataset object
    (DeviceArrays) and minibatches are gathered on the device in the order `data.batches` would produce (same RNG use)."""
    dev = base.device
    arrays = DeviceArrays.of(dataset, dev)                  # (one copy per split, shared with eval_imu)
    todo = list(batch_indices(len(dataset), batch_size, shuffle, rng))
    W = 30 if lower_net is None else 45
    log = torch.zeros((len(todo), W), dtype=torch.float32, device=dev)
    frames = np.empty(len(todo), dtype=np.float64)
    scratch = base.__dict__.setdefault("_eval_scratch", {})
    lmap = scratch.get("lmap")
    if lmap is None:
        lmap = scratch["lmap"] = torch.tensor(list(base.cfg.lower_joint_map), dtype=torch.int32, device=dev)
    with torch.no_grad():
        for i, idx in enumerate(todo):
            b = arrays.gather(idx)                          # fresh copies: the nets transform `data` in place (Q1)
            B, T = b["data"].shape[0], b["data"].shape[1]
            F = B * T
            frames[i] = F
            key = (B, T, W)
            if key not in scratch:
                scratch[key] = (torch.zeros((6, B, 64), device=dev), torch.zeros((6, B, 64), device=dev),
                                torch.empty((F, 43 if lower_net is not None else 30), dtype=torch.float32, device=dev))
            h0, c0, E = scratch[key]
            tgt, body = b["target"], b["skl"]
            R, t = base.head_pose(imu_net, b["imu"], b["R_R0R"], tgt)
            up = upper_net(b["data"], h0, c0, body, R, t)[0]
            if lower_net is None:
                hip.call("pose_errors_upper", up.contiguous(), tgt, F, E)
                ops.colsum(E, log[i])
            else:
                lo, _ = lower_net(up.clone(), b["data"], h0, h0, h0, h0, body, R, t)     # x already transformed once (Q1)
                hip.call("pose_errors", up.contiguous(), lo.contiguous(), tgt, F, E)
                ops.colsum(E, log[i, :43])
                hip.call("l1_loss", lo.contiguous(), tgt, lmap, 8, 21, F, 1.0, log[i, 43:45], None)
    out = log.cpu().numpy().astype(np.float64) / frames[:, None]             # the epoch's one device -> host read
    blocks.seq_xcd_raise()          # (the frozen IMU_Net's persistent launches: invalid head poses must not become a reported metric)
    return out


def evaluate_full(base, imu_net, upper_net, lower_net, dataset, batch_size, shuffle, rng=None):
    """IMU -> Upper -> Lower over a data split (reference Processor/Test/Demo_test.py:71-184, Train_Lower.py:232-332); returns
    (reference-style tuple, summary dict).  Device-resident split, one host read per pass (_epoch_eval)."""
    cfg = base.cfg
    m = _epoch_eval(base, dataset, batch_size, shuffle, rng, imu_net, upper_net, lower_net)
    eval_loss = float(np.mean(m[:, 43]))
    accu_l, accu_up, accu_lo = m[:, :21].mean(axis=1), m[:, 41], m[:, 42]
    accu_ll, angle_ll = m[:, :21].mean(axis=0), m[:, 21:41].mean(axis=0)
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
        out, s = evaluate_full(self, self.model_IMU, self.Upper_net, self.model, self.vis_data, 1, False)
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
        """Reference Processor/Train/Train_IMU.py:151-185.  Device-resident test split, per-minibatch figures kept in a device log
        (total loss from mmego_imu_loss; the position part = sum of |t - head joint| from mmego_l1_loss's distance output), one
        host read per epoch."""
        self.model_IMU.eval()
        dev = self.device
        self._test_dev = DeviceArrays.of(self.test_data, dev)
        if getattr(self, "_head_map", None) is None:
            self._head_map = torch.tensor([20], dtype=torch.int32, device=dev)
        todo = list(batch_indices(len(self.test_data), self.batchsize, True, self._rng))
        log = torch.zeros((len(todo), 3), dtype=torch.float32, device=dev)
        frames = np.empty(len(todo), dtype=np.float64)
        with torch.no_grad():
            for i, idx in enumerate(todo):
                b = self._test_dev.gather(idx)
                B, T = b["imu"].shape[0], b["imu"].shape[1]
                F = B * T
                frames[i] = F
                head = torch.empty((B, T, 3), dtype=torch.float32, device=dev)
                ops.copy2d(b["target"].view(F, 63)[:, 60:63], head.view(F, 3))
                R, t = self.model_IMU(b["imu"])
                hip.call("imu_loss", R.contiguous(), t.contiguous(), b["R_R0R"], head, F, 1.0, log[i, 0:1], None, None)
                hip.call("l1_loss", t.contiguous(), b["target"], self._head_map, 1, 21, F, 1.0, log[i, 1:3], None)
        m = log.cpu().numpy().astype(np.float64) / frames[:, None]          # the epoch's one device -> host read
        blocks.seq_xcd_raise()
        tot, pos = m[:, 0], m[:, 2]
        return float(np.mean(tot)), np.mean(np.stack((tot - 100.0 * pos, pos), axis=1), axis=0)

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
