"""IMUNet / UpperNet / UpperNetwlocal / LowerNet restated on CPU (oracle).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Plain torch fp32.  The
modules keep the reference's parameter names, shapes and construction order,
so a reference ``state_dict`` loads unchanged and ``torch.manual_seed`` gives
the same initial weights; the arithmetic is written channels-last
(rows x channels), which is also the layout of the HIP path.

Reference quirks reproduced on purpose: Q1 in-place
head transform, Q2 body row n % B, Q4 FK walk order, Q5 dead-point mask, Q6
degenerate fusion softmax, Q7 unused fc3 / ignored h0 arguments, and Q8 (found
while restating Net/GCN.py:351-353): the ST-GCN output (B,64,T,V) is
re-viewed, not permuted, as (B,T,V,64).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import geometry as geo
from . import graph as gr
from . import skeleton as sk


def _pw(conv, x):
    """k=1 Conv1d / 1x1 Conv2d as a row-wise linear map on channels-last x."""
    w = conv.weight.reshape(conv.weight.shape[0], -1)
    return F.linear(x, w, conv.bias)


def _bn(bn, x):
    """BatchNorm over all leading dims of channels-last x (train: batch stats)."""
    shape = x.shape
    y = F.batch_norm(x.reshape(-1, shape[-1]), bn.running_mean, bn.running_var, bn.weight, bn.bias,
                     bn.training, bn.momentum, bn.eps)
    if bn.training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1
    return y.view(shape)


class _Mlp3(nn.Module):
    """conv1/cb1 .. conv3/cb3: three (k=1 conv, BatchNorm1d, ReLU) stages."""

    def __init__(self, dims):
        super().__init__()
        self.conv1 = nn.Conv1d(dims[0], dims[1], 1)
        self.cb1 = nn.BatchNorm1d(dims[1])
        self.conv2 = nn.Conv1d(dims[1], dims[2], 1)
        self.cb2 = nn.BatchNorm1d(dims[2])
        self.conv3 = nn.Conv1d(dims[2], dims[3], 1)
        self.cb3 = nn.BatchNorm1d(dims[3])

    def stages(self, x):
        for conv, bn in ((self.conv1, self.cb1), (self.conv2, self.cb2), (self.conv3, self.cb3)):
            x = F.relu(_bn(bn, _pw(conv, x)))
        return x


class PointNet(_Mlp3):
    """6->8->16->24, out = cat(in[..., :4], feat).  Upper_Net.py:242-268."""

    def __init__(self):
        super().__init__((6, 8, 16, 24))

    def forward(self, pts):
        return torch.cat((pts[..., :4], self.stages(pts)), dim=-1)


class BasePointNet(_Mlp3):
    """6->16->32->61, out = cat(xyz, feat).  Lower_Net.py:40-72."""

    def __init__(self, hidden_dim=64):
        super().__init__((6, 16, 32, hidden_dim - 3))

    def forward(self, pts):
        return torch.cat((pts[..., :3], self.stages(pts)), dim=-1)


class _AttnPoolNet(_Mlp3):
    """MLP + Linear(64,1) softmax over the points axis + weighted sum."""

    def __init__(self, cin):
        super().__init__((cin, 32, 48, 64))
        self.attn = nn.Linear(64, 1)

    def forward(self, x):                       # (G, P, cin)
        f = self.stages(x)
        w = torch.softmax(self.attn(f), dim=1)  # over the P points of each group
        return (f * w).sum(dim=1), w


class GlobalPointNet(_AttnPoolNet):
    """Upper_Net.py:271-301."""

    def __init__(self):
        super().__init__(28)


class LocalPointNet(_AttnPoolNet):
    """Upper_Net.py:147-177 (31 = 3 anchor + 3 offset + 25 features)."""

    def __init__(self):
        super().__init__(31)


class GlobalModule(nn.Module):
    """Upper_Net.py:329-339."""

    def __init__(self):
        super().__init__()
        self.gpointnet = GlobalPointNet()
        self.grnn = nn.LSTM(64, 64, num_layers=3, batch_first=True, dropout=0.1, bidirectional=True)

    def forward(self, feats, h0, c0, B, T):
        vec, w = self.gpointnet(feats)
        seq, (hn, cn) = self.grnn(vec.view(B, T, -1), (h0, c0))
        return seq, w, hn, cn


def _six_d_head(x, B, T, joints):
    """Split (B,T,6*joints+extra) into rotations (B,T,joints,3,3) and the tail."""
    six = x[:, :, :6 * joints].reshape(B * T * joints, 6)
    return geo.rot6d_normalize(six).view(B, T, joints, 3, 3)


class MLPHead(nn.Module):
    """Upper_Net.py:343-364."""

    def __init__(self):
        super().__init__()
        self.fc1 = nn.Linear(128, 128)
        self.fc2 = nn.Linear(128, 14 * 6 + 3)

    def forward(self, seq, B, T):
        x = self.fc2(F.relu(self.fc1(seq)))
        return _six_d_head(x, B, T, 14), x[:, :, -3:]


class CombineModule(nn.Module):
    """Upper_Net.py:304-326."""

    def __init__(self):
        super().__init__()
        self.fc1 = nn.Linear(256, 128)
        self.fc2 = nn.Linear(128, 14 * 6 + 3)

    def forward(self, g, a, B, T):
        x = self.fc2(F.relu(self.fc1(torch.cat((g, a), dim=-1))))
        return _six_d_head(x, B, T, 14), x[:, :, -3:]


class LocalVoxelNet(nn.Module):
    """Conv3d(64,96,k=3) over the 3x3x3 anchor grid, then two 1x1x1 convs.

    Upper_Net.py:180-205.  With no padding the k=3 conv sees the whole grid, so
    it is one 1728->96 linear map whose input is ordered (cin, z, y, x).
    """

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv3d(64, 96, 3)
        self.cb1 = nn.BatchNorm3d(96)
        self.conv2 = nn.Conv3d(96, 128, 1)
        self.cb2 = nn.BatchNorm3d(128)
        self.conv3 = nn.Conv3d(128, 64, 1)
        self.cb3 = nn.BatchNorm3d(64)

    def forward(self, vox):                     # (F, 27, 64), anchors ordered [z][y][x]
        x = vox.transpose(1, 2).reshape(vox.shape[0], 64 * 27)
        for conv, bn in ((self.conv1, self.cb1), (self.conv2, self.cb2), (self.conv3, self.cb3)):
            x = F.relu(_bn(bn, _pw(conv, x)))
        return x


class LocalRNN(nn.Module):
    def __init__(self):
        super().__init__()
        self.rnn = nn.LSTM(64, 64, num_layers=3, batch_first=True, dropout=0.1, bidirectional=True)


class LocalModule(nn.Module):
    """Anchor grouping -> LocalPointNet -> LocalVoxelNet -> BiLSTM.  Upper_Net.py:217-239."""

    def __init__(self):
        super().__init__()
        self.apointnet = LocalPointNet()
        self.avoxel = LocalVoxelNet()
        self.arnn = LocalRNN()

    def forward(self, feats, h0, c0, B, T):
        grouped, idx = geo.anchor_grouping(feats[..., :3].contiguous(), feats[..., 3:].contiguous(), 8)
        self.last_group_idx = idx
        vox, w = self.apointnet(grouped.view(B * T * 27, 8, 31))
        vec = self.avoxel(vox.view(B * T, 27, 64))
        seq, (hn, cn) = self.arnn.rnn(vec.view(B, T, 64), (h0, c0))
        return seq, w, hn, cn


class UpperNet(nn.Module):
    """Upper_Net.py:367-388."""

    def __init__(self):
        super().__init__()
        self.module0 = PointNet()
        self.module1 = GlobalModule()
        self.mlpHead = MLPHead()

    def forward(self, x, h0_g, c0_g, initial_body, R, t):
        B, T = x.shape[0], x.shape[1]
        pts = geo.transform_to_head_(x, R, t)                 # mutates x (Q1)
        feats = self.module0(pts)
        seq, weights, hn, cn = self.module1(feats, h0_g, c0_g, B, T)
        q, head = self.mlpHead(seq, B, T)
        joints = geo.transform_to_world(geo.fk_upper(q, initial_body, head), R, t)
        return joints, q, weights, hn, cn


class UpperNetwlocal(nn.Module):
    """Upper_Net.py:406-432."""

    def __init__(self):
        super().__init__()
        self.module0 = PointNet()
        self.module1 = GlobalModule()
        self.module2 = LocalModule()
        self.module3 = CombineModule()

    def forward(self, x, h0_g, c0_g, h0_a, c0_a, initial_body, R, t):
        B, T = x.shape[0], x.shape[1]
        pts = geo.transform_to_head_(x, R, t)
        feats = self.module0(pts)
        g, gw, hn_g, cn_g = self.module1(feats, h0_g, c0_g, B, T)
        a, aw, hn_a, cn_a = self.module2(feats, h0_a, c0_a, B, T)
        q, head = self.module3(g, a, B, T)
        joints = geo.transform_to_world(geo.fk_upper(q, initial_body, head), R, t)
        return joints, q, gw, aw, hn_g, cn_g, hn_a, cn_a


# ----------------------------------------------------------------------------
# ST-GCN over the 15 upper joints -- Net/GCN.py
# ----------------------------------------------------------------------------

class _GraphConv(nn.Module):
    def __init__(self, cin, cout, K):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout * K, kernel_size=(1, 1))


class StGcnBlock(nn.Module):
    """GCN.py:67-147 on channels-last x (B,T,V,C)."""

    def __init__(self, cin, cout, K, taps=9):
        super().__init__()
        self.K, self.cout, self.taps = K, cout, taps
        self.gcn = _GraphConv(cin, cout, K)
        self.tcn = nn.ModuleDict({"0": nn.BatchNorm2d(cout),
                                  "2": nn.Conv2d(cout, cout, (taps, 1), (1, 1), (taps // 2, 0)),
                                  "3": nn.BatchNorm2d(cout)})
        self.residual = nn.ModuleDict({"0": nn.Conv2d(cin, cout, kernel_size=1, stride=(1, 1)),
                                       "1": nn.BatchNorm2d(cout)})

    def forward(self, x, A):                    # x (B,T,V,Cin), A (K,V,V)
        B, T, V, _ = x.shape
        res = _bn(self.residual["1"], _pw(self.residual["0"], x))
        z = _pw(self.gcn.conv, x).view(B, T, V, self.K, self.cout)
        y = torch.einsum("btvkc,kvw->btwc", z, A)              # GCN.py:62
        y = F.relu(_bn(self.tcn["0"], y))
        half = self.taps // 2
        ypad = F.pad(y, (0, 0, 0, 0, half, half))              # zero-pad T
        w = self.tcn["2"].weight[:, :, :, 0]                   # (Cout, Cin, taps)
        out = self.tcn["2"].bias.view(1, 1, 1, -1).expand(B, T, V, -1)
        for tap in range(self.taps):
            out = out + F.linear(ypad[:, tap:tap + T], w[:, :, tap])
        out = _bn(self.tcn["3"], out)
        return F.relu(out + res)


class GcnModel(nn.Module):
    """GCN.py:281-355 (extract_feature path only)."""

    def __init__(self, in_channels=3, hidden_dim=64, strategy="distance"):
        super().__init__()
        A = torch.tensor(gr.adjacency(strategy), dtype=torch.float32)
        self.register_buffer("A", A)
        K = A.shape[0]
        self.data_bn = nn.BatchNorm1d(in_channels * A.shape[1])
        self.gcn_networks = nn.ModuleList((StGcnBlock(in_channels, 32, K), StGcnBlock(32, 64, K),
                                           StGcnBlock(64, 128, K)))
        self.edge_importance = nn.ParameterList([nn.Parameter(torch.ones(A.shape)) for _ in range(3)])
        self.fcn = nn.Conv2d(128, hidden_dim, kernel_size=1)

    def extract_feature(self, joints):          # (B,T,V,3) channels-last
        B, T, V, C = joints.shape
        x = _bn(self.data_bn, joints.reshape(B, T, V * C)).view(B, T, V, C)   # channel = v*3+c
        for blk, imp in zip(self.gcn_networks, self.edge_importance):
            x = blk(x, self.A * imp)
        y = _pw(self.fcn, x)                                    # (B,T,V,64)
        # Q8: reference holds this as (B,64,T,V) and re-views the memory as (B,T,V,64)
        return y.permute(0, 3, 1, 2).contiguous().view(B, T, V, -1)


class KeyEncoder(nn.Module):
    def __init__(self, hidden_dim=64):
        super().__init__()
        self.gcn = GcnModel(3, hidden_dim, "distance")


class PointEncoder(nn.Module):
    def __init__(self, hidden_dim=64):
        super().__init__()
        self.module0 = BasePointNet(hidden_dim)


class FusionModule(nn.Module):
    """Lower_Net.py:75-136."""

    def __init__(self, hidden_dim=64):
        super().__init__()
        self.fc0 = nn.Linear(hidden_dim * 2 + sk.JOINTS_UPPER * 3, 128)
        self.fc1 = nn.Linear(128, 64)
        self.to_q = nn.Linear(hidden_dim, hidden_dim)
        self.to_k = nn.Linear(hidden_dim, hidden_dim)
        self.to_v = nn.Linear(hidden_dim, hidden_dim)
        self.scale = hidden_dim ** -0.5
        self.fc2 = nn.Linear(64, 6 * 6 + 2 * 3)
        self.attn = nn.Linear(hidden_dim * 2, 1)
        self.rnn_pk = nn.LSTM(hidden_dim * 3, hidden_dim, num_layers=3, batch_first=True, dropout=0.1,
                              bidirectional=True)

    def forward(self, p_vec, k_vec, upper, B, T):
        # p_vec (BT,64,64) k_vec (BT,15,64) upper (B,T,15,3)
        scores = (self.to_q(p_vec) @ self.to_k(k_vec).transpose(-2, -1)) * self.scale
        mixed = torch.softmax(scores, dim=-1) @ self.to_v(k_vec)
        both = torch.cat((p_vec, mixed), dim=-1)                # (BT,64,128)
        gate = torch.softmax(self.attn(both), dim=-1)           # Q6: size-1 axis -> all ones
        a_vec = (both * gate).sum(dim=1).view(B, T, -1)
        k_mean = k_vec.mean(dim=1).view(B, T, -1)
        seq, _ = self.rnn_pk(torch.cat((a_vec, k_mean), dim=-1))
        x = torch.cat((seq, upper.reshape(B, T, -1)), dim=-1)
        x = self.fc2(F.relu(self.fc1(F.relu(self.fc0(x)))))
        return _six_d_head(x, B, T, 6), x[:, :, -6:-3], x[:, :, -3:]


class LowerNet(nn.Module):
    """Lower_Net.py:170-239."""

    def __init__(self, hidden_dim=64):
        super().__init__()
        self.pointEncoder = PointEncoder(hidden_dim)
        self.keyEncoder = KeyEncoder(hidden_dim)
        self.fusion = FusionModule(hidden_dim)

    def forward(self, upper_l, x, h0_p, c0_p, h0_k, c0_k, initial_body, R, t, pin_select_idx=None):
        """``pin_select_idx`` (tests only): use these top-64 indices instead of sorting, to replay the
        reference's build-specific tie order when checking against its golden outputs."""
        B, T = x.shape[0], x.shape[1]
        pts = geo.transform_to_head_(x, R, t)                   # second transform when Upper ran first (Q1)
        upper = geo.transform_to_head_(upper_l.reshape(B, T, sk.JOINTS_UPPER, 3).clone(), R, t)
        upper = upper.view(B, T, sk.JOINTS_UPPER, 3)
        low_pts, idx = geo.top_x_select(pts)
        self.last_select_idx = idx
        if pin_select_idx is not None:
            low_pts = pts[torch.arange(B * T).view(-1, 1), pin_select_idx]
        p_vec = self.pointEncoder.module0(low_pts)
        k_vec = self.keyEncoder.gcn.extract_feature(upper).reshape(B * T, sk.JOINTS_UPPER, -1)
        self.last_p_vec, self.last_k_vec = p_vec, k_vec
        q, hip_l, hip_r = self.fusion(p_vec, k_vec, upper, B, T)
        joints = geo.transform_to_world(geo.fk_lower(q, hip_l, hip_r, initial_body), R, t)
        return joints, q


class IMUNet(nn.Module):
    """IMU_Net.py:50-94."""

    def __init__(self, input_n=15, output_n=9, hidden_n=512, n_rnn_layer=2, bidirectional=True, dropout=0):
        super().__init__()
        d = 2 if bidirectional else 1
        self.fc1 = nn.Linear(input_n, hidden_n)
        self.fc2 = nn.Linear(hidden_n * d, output_n)
        self.fc3 = nn.Linear(output_n, 3)          # Q7: in the state_dict, never used
        self.rnn_fast = nn.LSTM(hidden_n, hidden_n, n_rnn_layer, bidirectional=bidirectional, batch_first=True,
                                dropout=dropout)
        self.rnn_slow = nn.LSTM(2 * hidden_n, hidden_n, n_rnn_layer, bidirectional=bidirectional,
                                batch_first=True, dropout=dropout)
        self.attn = nn.Linear(hidden_n * d, 1)

    def forward(self, imu, h0_i=None):
        B, T, S, _ = imu.shape
        fast, _ = self.rnn_fast(F.relu(self.fc1(imu.reshape(B * T, S, -1))), h0_i)
        w = torch.softmax(self.attn(fast), dim=1)
        pooled = (fast * w).sum(dim=1).view(B, T, -1)
        slow, _ = self.rnn_slow(pooled, h0_i)
        out = self.fc2(slow).reshape(B * T, -1)
        R = geo.rot6d_imu(out[:, :6]).view(B, T, 3, 3)
        return R, out[:, 6:].reshape(B, T, 3)
