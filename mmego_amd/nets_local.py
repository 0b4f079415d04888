"""UpperNetwlocal on the HIP path: UpperNet plus the anchor ("voxel") branch.

Reference Net/Upper_Net.py:406-432 (defined there but constructed by no trainer; part of the drop-in surface):
PointNet -> {GlobalModule, LocalModule (anchor grouping -> LocalPointNet -> LocalVoxelNet -> BiLSTM)} ->
CombineModule -> FK -> Transform2R.  forward returns the reference's 8-tuple.
"""
import os

import torch
import torch.nn as nn

from . import blocks, hip, ops
from .blocks import LstmParams
from .nets import GlobalModule, PointNet, _Bridge, _Mlp3, _NetBase, _f32c, _require_gpu

N_ANCHOR, N_GROUP = 27, 8
_LOCAL_FUSED = True      # the anchor branch on the fused kernels of local.hip
_VOX_FUSED = True        # LocalVoxelNet's training step on the kernels of vox.hip (tests flip it: the generic launches)
_LOCAL_NWG = 256                                                     # workgroups of mmego_local_group_l1 (= BatchNorm partial records)


def anchor_grid():
    """(27,3) fp32 anchors in [z][y][x] order: x in {0,.3,.6}, y,z in {-.3,0,.3} (Upper_Net.py:75-97)."""
    pts = [(0 + xi * 0.3, -0.3 + yi * 0.3, -0.3 + zi * 0.3) for zi in range(3) for yi in range(3) for xi in range(3)]
    return torch.tensor(pts, dtype=torch.float32)


class LocalPointNet(_Mlp3):
    def __init__(self):
        super().__init__((24 + 4 + 3, 32, 48, 64))
        self.attn = nn.Linear(64, 1)


class LocalVoxelNet(nn.Module):
    """Conv3d(64,96,k=3) over the whole 3x3x3 grid (= one 1728->96 map) and two 1x1x1 convs (Upper_Net.py:180-205)."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv3d(64, 96, kernel_size=(3, 3, 3), padding=(0, 0, 0))
        self.cb1 = nn.BatchNorm3d(96)
        self.conv2 = nn.Conv3d(96, 128, kernel_size=(1, 1, 1))
        self.cb2 = nn.BatchNorm3d(128)
        self.conv3 = nn.Conv3d(128, 64, kernel_size=(1, 1, 1))
        self.cb3 = nn.BatchNorm3d(64)


class LocalRNN(nn.Module):
    def __init__(self):
        super().__init__()
        self.rnn = LstmParams(64, 64, 3, dropout=0.1, bidirectional=True)


class LocalModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.template_point = anchor_grid().view(3, 3, 3, 3)      # plain tensor attribute, as in the reference
        self.apointnet = LocalPointNet()
        self.avoxel = LocalVoxelNet()
        self.arnn = LocalRNN()


class CombineModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.fc1 = nn.Linear(256, 128)
        self.fc2 = nn.Linear(128, 14 * 6 + 3)


class UpperNetwlocal(_NetBase):
    """forward(x, h0_g, c0_g, h0_a, c0_a, initial_body, R, t) ->
    (l, q, global_weights, anchor_weights, hn_g, cn_g, hn_a, cn_a).  MUTATES x (Q1)."""

    def __init__(self):
        super().__init__()
        self.module0 = PointNet()
        self.module1 = GlobalModule()
        self.module2 = LocalModule()
        self.module3 = CombineModule()
        self._anchors = None

    def anchors(self, dev):
        if self._anchors is None or self._anchors.device != dev:
            self._anchors = self.module2.template_point.reshape(N_ANCHOR, 3).to(dev).contiguous()
        return self._anchors

    def _local_fusable(self, N):
        lp = self.module2.apointnet
        dims = [tuple(c.weight.shape[:2]) for c in (lp.conv1, lp.conv2, lp.conv3)]
        return bool(_LOCAL_FUSED and N in (64, 128, 256) and dims == [(32, 31), (48, 32), (64, 48)] and lp.attn.weight.shape == (1, 64))

    def _local_table(self):
        """Host-side pointer table of mmego_local_front_eval (20 device pointers), rebuilt when a tensor moved."""
        lp = self.module2.apointnet
        ts = [v for conv, bn in ((lp.conv1, lp.cb1), (lp.conv2, lp.cb2), (lp.conv3, lp.cb3))
              for v in (conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var)] + [lp.attn.weight, lp.attn.bias]
        ptrs = tuple(v.data_ptr() for v in ts)
        ent = self.__dict__.get("_local_tab")
        if ent is None or ent[0] != ptrs:
            ent = self.__dict__["_local_tab"] = (ptrs, torch.tensor(ptrs, dtype=torch.int64))
        return ent[1]

    def local_branch_bytes(self, F, N):
        """HBM bytes per frame the anchor branch's activations move between launches in one training step (written once + read
        once per consumer; parameters and the shared per-point features excluded): the fused path against the launch chain."""
        g = N_ANCHOR * N_GROUP
        fused = 4 * (g * (32 + 48 + 64) * 3         # z1..z3: written, read by the next stage, read by backward (the gathered rows
                                                    # themselves never reach memory: LDS forward, re-gathered backward)
                     + g * (64 + 48 + 32 + 31) * 2  # dl3, dy2, dy1, dgrouped
                     + N_ANCHOR * 64 * 4 + g * 2) + 8 * g
        chain = fused + 4 * (g * 31 * 3             # gathered rows written, read by the first layer and by its weight gradient
                             + g * 64 * 4           # l3 written, read by pooling, by its backward, by the BatchNorm backward
                             + g * 64 * 2           # dl3 read by the separate BatchNorm reduce
                             + N_ANCHOR * 64 * 4)   # vox / dvox through the transposes
        return {"fused": fused, "launch_chain": chain, "gathered_rows": 4 * g * 31, "what": self.local_branch_bytes.__doc__.split(":")[0]}

    def forward(self, x, h0_g, c0_g, h0_a, c0_a, initial_body, R, t):
        _require_gpu(x, "UpperNetwlocal")
        args = (x, h0_g, c0_g, h0_a, c0_a, initial_body, R, t)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            self.flat()
            return _Bridge.apply(self, 1, args, *self._flat.params)
        return self._forward_impl(*args, stash=False)

    def _forward_impl(self, x, h0g, c0g, h0a, c0a, body, R, t, stash=True, x_src=None):
        self.flat()
        training = self.training
        ar = self.arena("train" if stash else "eval")
        if not (x.dtype == torch.float32 and x.is_contiguous()):
            raise ValueError("UpperNetwlocal: x must be a contiguous fp32 tensor (it is transformed in place)")
        B, T, N, Cx = x.shape
        F, rows = B * T, B * T * N
        dev = x.device
        R, t, body = _f32c(R), _f32c(t), _f32c(body)
        h0g, c0g, h0a, c0a = [_f32c(v) if v is not None else None for v in (h0g, c0g, h0a, c0a)]
        feats = ar.get("feats", (rows, 28))
        keep = ar.get("pts", (rows, Cx)) if stash else None
        if Cx <= 8:
            # Q1: in place on the caller's tensor; the copy kept for backward and the xyz + intensity columns of the feature buffer leave
            # from the same launch, and a trainer's fresh minibatch (x_src) enters through it (as in nets.UpperNet)
            ops.transform2h_(x, R, t, src=x_src, keep=keep, feats=feats, nfeat=4)
            pts = keep if stash else x.view(rows, Cx)
        else:
            if x_src is not None:
                ops.copy2d(x_src.view(rows, Cx), x.view(rows, Cx))
            ops.transform2h_(x, R, t)
            pts = x.view(rows, Cx)
            if stash:
                ops.copy2d(pts, keep)
                pts = keep
            ops.copy2d(pts[:, :4], feats[:, :4])
        blocks.mlp3_forward(ar, "m0", self.module0, pts, feats[:, 4:28], training)
        # global branch
        g3 = ar.get("g3", (rows, 64))
        vec = ar.get("vec", (F, 64))
        gw = torch.empty((F, N, 1), dtype=torch.float32, device=dev)
        gpn = self.module1.gpointnet
        self._gpool_fused = bool(stash and blocks.pool128_fusable(gpn, feats, N, training))
        if self._gpool_fused:       # the pooling inside the chain's last launch: g3 (the activated rows) is never stored
            blocks.mlp3_forward(ar, "gp", gpn, feats, g3, training, pool=(gpn.attn, vec, gw))
        else:
            blocks.mlp3_forward(ar, "gp", gpn, feats, g3, training)
            blocks.attn_pool_forward(g3, gpn.attn, F, N, 64, vec, gw)
        p_g = self._drop_p(self.module1.grnn) if stash else 0.0
        cat = ar.get("cat", (F, 256))
        # local branch: grouping -> LocalPointNet (+attention pool over the 8 members) -> voxel net -> BiLSTM
        grows = F * N_ANCHOR * N_GROUP
        gidx = ar.get("gidx", (F, N_ANCHOR, N_GROUP), dtype=torch.int64)
        self.last_group_idx = gidx
        lp = self.module2.apointnet
        voxT = ar.get("voxT", (F, 64 * N_ANCHOR))
        aw = torch.empty((F * N_ANCHOR, N_GROUP, 1), dtype=torch.float32, device=dev)
        fused = self._local_fusable(N)
        self._local_was_fused = bool(fused and training and stash)      # (read by backward only: as _gpool_fused)
        if fused and training:
            # grouping with LocalPointNet's first conv behind it in the same kernel (the gathered rows go from LDS into the product and
            # are never stored: the layer's backward gathers them again through the indices), two fused layer launches, then BatchNorm + ReLU + the 8-way
            # softmax pooling in one kernel that writes the pooled vectors in the Conv3d input order: the activated 64-channel
            # tensor, the pooling pass over it and the transpose launch do not exist (local.hip)
            nblk = hip.lib().mmego_mlp_train_nblk(grows)
            nwg = min(_LOCAL_NWG, F)
            grouped = None           # (not kept: the first layer's weight gradient gathers its input again through the indices)
            z1, z2, z3 = ar.get("lp.z1", (grows, 32)), ar.get("lp.z2", (grows, 48)), ar.get("lp.z3", (grows, 64))
            part1 = ar.get("lp.sp1g", (nwg * 2 * 64,), dtype=torch.float64)
            part2, part3 = (ar.get("lp.sp%d" % i, (nblk * 2 * 64,), dtype=torch.float64) for i in (2, 3))
            st1, st2, st3 = (ops.BnState(ar, "lp.bn%d" % i, c) for i, c in ((1, 32), (2, 48), (3, 64)))
            hip.call("local_group_l1", feats, 28, F, N, 25, self.anchors(dev), gidx, grouped, lp.conv1.weight, lp.conv1.bias, 32, z1, 32,
                     part1, nwg, None)
            hip.call("mlp_fwd_layer_n", z1, 32, grows, 32, part1, nwg, lp.cb1.weight, lp.cb1.bias, float(lp.cb1.eps), lp.cb1.running_mean,
                     lp.cb1.running_var, float(lp.cb1.momentum), st1.all, lp.conv2.weight, lp.conv2.bias, 48, z2, 48, part2)
            hip.call("mlp_fwd_layer", z2, 48, grows, 48, part2, lp.cb2.weight, lp.cb2.bias, float(lp.cb2.eps), lp.cb2.running_mean,
                     lp.cb2.running_var, float(lp.cb2.momentum), st2.all, lp.conv3.weight, lp.conv3.bias, 64, z3, 64, part3)
            hip.call("pool8_bn_act", z3, 64, grows, part3, lp.cb3.weight, lp.cb3.bias, float(lp.cb3.eps), lp.cb3.running_mean,
                     lp.cb3.running_var, float(lp.cb3.momentum), st3.all, lp.attn.weight, lp.attn.bias, voxT, aw)
        elif fused and not training and len({float(bn.eps) for bn in (lp.cb1, lp.cb2, lp.cb3)}) == 1:
            # eval: grouping, the three conv + BatchNorm(running statistics, folded) + ReLU stages and the pooling in ONE launch; neither
            # the gathered rows nor any per-member activation reaches HBM
            hip.call("local_front_eval", feats, 28, F, N, 25, self.anchors(dev), gidx, self._local_table(), float(lp.cb1.eps), voxT, aw)
        else:
            grouped = ar.get("grouped", (grows, 31))
            if fused:           # (eval mode with BatchNorms of unequal eps -- local_front_eval folds with ONE eps: the wave-parallel grouping alone)
                hip.call("local_group_l1", feats, 28, F, N, 25, self.anchors(dev), gidx, grouped, None, None, 0, None, 0, None, _LOCAL_NWG, None)
            else:
                hip.call("anchor_group", feats, 28, F, N, 25, self.anchors(dev), gidx, grouped, None)
            l3 = ar.get("l3", (grows, 64))
            blocks.mlp3_forward(ar, "lp", lp, grouped, l3, training)
            vox = ar.get("vox", (F * N_ANCHOR, 64))
            blocks.attn_pool_forward(l3, lp.attn, F * N_ANCHOR, N_GROUP, 64, vox, aw)
            hip.call("transpose_batched", vox, voxT, F, N_ANCHOR, 64)     # (F,27,64) -> (F,64,27): conv input order (cin,z,y,x)
        vvec = ar.get("vvec", (F, 64))
        self._vox_fused = bool(_VOX_FUSED and stash and training and blocks.vox_fusable(self.module2.avoxel, voxT))
        if self._vox_fused:         # LocalVoxelNet's 512-row chain on its own kernels: 4 launches instead of 12 (vox.hip)
            blocks.voxel_forward(ar, "vx", self.module2.avoxel, voxT, vvec)
        else:
            blocks.mlp3_forward(ar, "vx", self.module2.avoxel, voxT, vvec, training)
        p_a = self._drop_p(self.module2.arnn.rnn) if stash else 0.0
        # the global and the anchor BiLSTM(64) stacks (Net/Upper_Net.py:333-339, :208-216: same shape, nothing in common) layer by layer side
        # by side: one sequence-kernel launch per layer for both (blocks.lstm64_forward_multi)
        (seq_g, hn_g, cn_g), (seq_a, hn_a, cn_a) = blocks.lstm64_forward_multi(
            ar, [("grnn", self.module1.grnn, vec, h0g, c0g, p_g, self.seed_counter(), 0),
                 ("arnn", self.module2.arnn.rnn, vvec, h0a, c0a, p_a, self.seed_counter(), 1)], B, T, stash,
            last_out=(cat[:, :128], cat[:, 128:]))          # (the last layers write the two halves of the concatenation themselves)
        # combine head
        h1 = ar.get("h1", (F, 128))
        ops.linear(cat, self.module3.fc1.weight, self.module3.fc1.bias, h1, relu=True)
        y = ar.get("y", (F, 87))
        ops.linear(h1, self.module3.fc2.weight, self.module3.fc2.bias, y)
        q = torch.empty((B, T, 14, 3, 3), dtype=torch.float32, device=dev)
        jh = ar.get("jh", (F, 15, 3))
        l = torch.empty((B, T, 15, 3), dtype=torch.float32, device=dev)
        tick = self._flat.tick_args(self.seed_counter()) if training else (None, 0, None)   # BatchNorm counters + dropout seed
        self._head_fk(ar, 0, y, body, B, F, q, jh, R, t, l, tick, stash)    # kinematics + head-to-world transform (+ loss), one launch
        if stash:
            self._saved = (B, T, N, R, body, c0g, c0a, gw, aw)
        return l, q, gw, aw, hn_g, cn_g, hn_a, cn_a

    def _backward_impl(self, dl):
        ar = self.arena("train")
        B, T, N, R, body, c0g, c0a, gw, aw = self._saved
        F, rows = B * T, B * T * N
        grows = F * N_ANCHOR * N_GROUP
        G = self._flat.grad
        dl = _f32c(dl)
        y, h1, cat = ar.get("y", (F, 87)), ar.get("h1", (F, 128)), ar.get("cat", (F, 256))
        dy = ar.get("dy", (F, 87))
        if not getattr(self, "_dy_ready", False):
            hip.call("head_fk_backward", 0, y, body, B, F, dl, dy, R)           # (world -> head frame inside the kernel)
        dh1 = ar.get("dh1", (F, 128))
        blocks.linear_backward(dy, h1, self.module3.fc2, G, dh1, relu_input=True)
        dcat = ar.get("dcat", (F, 256))
        blocks.linear_backward(dh1, cat, self.module3.fc1, G, dcat)
        with blocks.dw_reduce_group():                         # the three chains' weight-gradient partials: one reduce launch
            # global branch
            vec = ar.get("vec", (F, 64))
            vvec = ar.get("vvec", (F, 64))
            dvec, dvvec = blocks.lstm64_backward_multi(
                ar, [("grnn", self.module1.grnn, vec, c0g, dcat[:, :128], self._drop_p(self.module1.grnn)),
                     ("arnn", self.module2.arnn.rnn, vvec, c0a, dcat[:, 128:], self._drop_p(self.module2.arnn.rnn))], B, T, G, True)
            g3, dg3 = ar.get("g3", (rows, 64)), ar.get("dg3", (rows, 64))
            feats = ar.get("feats", (rows, 28))
            gpn = self.module1.gpointnet
            if getattr(self, "_gpool_fused", False):
                blocks.pool128_backward_fused(ar, "gp", gpn, gpn.attn, gw, vec, dvec, rows, dg3, G)
                dfeats = blocks._mlp3_backward_fused(ar, "gp", gpn, feats, dg3, G, True, have_sums=True)
            else:
                blocks.attn_pool_backward(ar, "gpool", g3, gpn.attn, gw, dvec, F, N, 64, dg3, G)
                dfeats = blocks.mlp3_backward(ar, "gp", gpn, feats, g3, dg3, G, True)
            # local branch
            voxT = ar.get("voxT", (F, 64 * N_ANCHOR))
            if getattr(self, "_vox_fused", False):
                dvoxT = blocks.voxel_backward(ar, "vx", self.module2.avoxel, voxT, vvec, dvvec, G)
            else:
                dvoxT = blocks.mlp3_backward(ar, "vx", self.module2.avoxel, voxT, vvec, dvvec, G, True)
            gidx = ar.get("gidx", (F, N_ANCHOR, N_GROUP), dtype=torch.int64)
            lp = self.module2.apointnet
            if getattr(self, "_local_was_fused", False):
                # pooling backward with the activated rows recomputed, the last stage's BatchNorm sums and the attention parameter
                # partials from the same kernel; the pooled gradient is read in the Conv3d order (no transpose launch)
                nblk = hip.lib().mmego_pool8_nblk(grows)
                z3, dl3 = ar.get("lp.z3", (grows, 64)), ar.get("dl3", (grows, 64))
                gp3 = ar.get("lp.gp3", (nblk * 2 * 64,), dtype=torch.float64)
                awp = ar.get("lpool.awp", (nblk, 128))
                hip.call("pool8_backward", z3, 64, grows, ops.BnState(ar, "lp.bn3", 64).all, aw, dvoxT, lp.attn.weight, dl3, 64, gp3, awp)
                gw, gb = G(lp.attn.weight).view(-1), G(lp.attn.bias)
                if nblk <= 256 and blocks._dw_group is not None:          # with the pass's other partial sums (blocks.dw_reduce_group)
                    blocks._dw_group.append(hip.DwRed(hip.ptr(awp), hip.ptr(gw), 1, 64, 0, nblk, 128))
                    blocks._dw_group.append(hip.DwRed(hip.ptr(awp[:, 64:]), hip.ptr(gb), 1, 1, 0, nblk, 128))
                elif gb.data_ptr() == gw.data_ptr() + 4 * gw.numel():     # weight and bias gradient slots back to back: one column sum
                    ops.colsum(awp[:, :65], torch.as_strided(gw, (65,), (1,)))
                else:
                    ops.colsum(awp[:, :64], gw)
                    ops.colsum(awp[:, 64:65], gb)
                dgrouped = blocks._mlp3_backward_fused(ar, "lp", lp, None, dl3, G, True, have_sums=True,
                                                       gather=(grows, gidx, feats, 28, self.anchors(feats.device), N, 25))
                hip.call("anchor_scatter", dgrouped, gidx, F, N, 25, dfeats, 28)
            else:
                grouped = ar.get("grouped", (grows, 31))
                dvox = ar.get("dvox", (F * N_ANCHOR, 64))
                hip.call("transpose_batched", dvoxT, dvox, F, 64, N_ANCHOR)       # (F,64,27) -> (F,27,64)
                l3, dl3 = ar.get("l3", (grows, 64)), ar.get("dl3", (grows, 64))
                blocks.attn_pool_backward(ar, "lpool", l3, lp.attn, aw, dvox, F * N_ANCHOR, N_GROUP, 64, dl3, G)
                dgrouped = blocks.mlp3_backward(ar, "lp", lp, grouped, l3, dl3, G, True)
                hip.call("anchor_group_backward", dgrouped, gidx, F, N, 25, dfeats, 28)
            pts = ar.get("pts", (rows, 6))
            blocks.mlp3_backward(ar, "m0", self.module0, pts, feats[:, 4:28], dfeats[:, 4:28], G, False)
