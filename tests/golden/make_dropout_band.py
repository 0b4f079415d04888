#!/usr/bin/env python3
"""G10: the reference's run-to-run band of a DROPOUT-ACTIVE training run.

Run once in the build container (the reference lives at /root/reference and never travels):
    python tests/golden/make_dropout_band.py
Only numbers are written (tests/golden/g10_dropout_band.npz); no reference source is copied.

What is run, on the REAL reference modules (Net/Upper_Net.py:367 UpperNet, Net/Lower_Net.py:170 LowerNet, with their
nn.LSTM(dropout=...) live, Upper_Net.py:333 / Lower_Net.py:91-93): the per-minibatch body of Processor/Train/Train_Upper.py:
134-187 / Train_Lower.py:155-230 (forward, L1Loss(sum), backward, Adam) with the recorded head pose (R_R0R, head joint),
200 steps on the 16 real sequences of tests/golden/real16.npz as ONE minibatch (B=16, T=20, N=128), lr 3e-4.  The
initial weights are fixed (torch.manual_seed(INIT) before construction); ONLY the dropout stream varies
(torch.manual_seed(dseed) after construction).  Statistic: mean per-joint Euclidean error in cm of the eval-mode forward on
the same 16 sequences after the 200 steps ("final train-set joint error"), plus the loss at steps 50/100/150/200.

  p = 0.1 (the reference's setting)  x 6 dropout seeds  -> the band
  p = 0.0                            x 1                -> deterministic 200-step curve (a long-horizon parity pin)
  p = 0.2                            x 3 dropout seeds  -> shows the statistic separates a wrong dropout rate from the band
  p = 0.0, initial weights x (1 + 1e-7 N(0,1)), 5 draws -> how far ROUNDING-SIZED differences move the end point of a
                                                           dropout-free run (200 Adam steps at lr 3e-4 amplify them): the
                                                           band a different-but-correct implementation can be held to
(`--chaos-only` adds the last block to an existing fixture without re-running the others.)

The same runs are repeated with the build's CPU oracle (oracle/nets.py) so that tests/test_oracle_golden.py can hold the
oracle to the reference without importing it.
"""
import os
import sys
import time
import types

import numpy as np

REF = os.environ.get("MMEGO_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))

sys.dont_write_bytecode = True
sys.path.insert(0, REF)
import matplotlib  # noqa: E402

matplotlib.use("Agg")
for _m in ("seaborn", "imageio", "imageio.v2"):      # plot-only deps of Utils.py, not installed
    sys.modules.setdefault(_m, types.ModuleType(_m))

import torch  # noqa: E402

torch.set_num_threads(8)

from Config.config import Config  # noqa: E402
from Net.Lower_Net import LowerNet  # noqa: E402
from Net.Upper_Net import UpperNet  # noqa: E402

STEPS, LR, INIT_UPPER, INIT_LOWER = 200, 3e-4, 1, 2
SEEDS_BAND = (10, 11, 12, 13, 14, 15)
SEEDS_P02 = (10, 11, 12)


def set_lstm_dropout(model, p):
    for m in model.modules():
        if isinstance(m, torch.nn.LSTM):
            m.dropout = p


def train_run(make_net, fwd, tgt, init_seed, dseed, p, perturb=0):
    torch.manual_seed(init_seed)
    net = make_net().train()
    set_lstm_dropout(net, p)
    if perturb:
        gen = torch.Generator().manual_seed(perturb)
        with torch.no_grad():
            for q in net.parameters():
                q.mul_(1 + 1e-7 * torch.randn(q.shape, generator=gen))
    torch.manual_seed(dseed)
    opt = torch.optim.Adam(net.parameters(), lr=LR)
    loss_fn = torch.nn.L1Loss(reduction="sum")
    curve = []
    for s in range(1, STEPS + 1):
        opt.zero_grad()
        loss = loss_fn(fwd(net), tgt)
        loss.backward()
        opt.step()
        if s % 50 == 0:
            curve.append(loss.item())
    net.eval()
    with torch.no_grad():
        err = torch.sqrt(torch.sum(torch.square(fwd(net) - tgt), dim=-1)).mean().item() * 100.0
    return err, curve


def main():
    real = np.load(os.path.join(OUT, "real16.npz"))
    x0, target, body, R = [torch.tensor(real[k]) for k in ("x", "target", "skl", "R")]
    t = target[:, :, 20].contiguous()
    B = x0.shape[0]
    h0, c0 = torch.zeros(6, B, 64), torch.zeros(6, B, 64)
    up_map, lo_map = list(Config.upper_joint_map), list(Config.lower_joint_map)
    # Lower stage: the frozen, pretrained Upper_Net's joints (Train_Lower.py:129-137), x as that forward leaves it (Q1)
    upf = UpperNet()
    upf.load_state_dict({k: torch.tensor(v) for k, v in np.load(os.path.join(OUT, "w_upper_pretrained.npz")).items()})
    upf.eval()
    with torch.no_grad():
        x_l = x0.clone()
        up_l = upf(x_l, h0, c0, body, R, t)[0].clone()

    sys.path.insert(0, ROOT)
    from oracle import nets as on

    sides = {"ref": (UpperNet, lambda: LowerNet(hidden_dim=64)), "oracle": (on.UpperNet, lambda: on.LowerNet(64))}
    out = {"steps": np.asarray(STEPS), "lr": np.asarray(LR), "init_upper": np.asarray(INIT_UPPER),
           "init_lower": np.asarray(INIT_LOWER), "seeds_band": np.asarray(SEEDS_BAND), "seeds_p02": np.asarray(SEEDS_P02)}
    path = os.path.join(OUT, "g10_dropout_band.npz")
    chaos_only = "--chaos-only" in sys.argv
    if chaos_only:
        out = dict(np.load(path))
    for side, (mk_up, mk_lo) in sides.items():
        stages = {"upper": (mk_up, lambda m: m(x0.clone(), h0, c0, body, R, t)[0], target[:, :, up_map], INIT_UPPER),
                  "lower": (mk_lo, lambda m: m(up_l.clone(), x_l.clone(), h0, c0, h0, c0, body, R, t)[0], target[:, :, lo_map], INIT_LOWER)}
        for stage, (mk, fwd, tgt, init) in stages.items():
            if side == "ref":
                t0 = time.time()
                runs = [train_run(mk, fwd, tgt, init, 10, 0.0, perturb=k) for k in (1, 2, 3, 4, 5)]
                out["ref.%s.p00_perturbed.err_cm" % stage] = np.asarray([r[0] for r in runs])
                out["ref.%s.p00_perturbed.loss_curve" % stage] = np.asarray([r[1] for r in runs])
                print("%-6s %-5s p00 perturbed 1e-7: err_cm %s  (%.0f s)" % (side, stage, np.round([r[0] for r in runs], 4), time.time() - t0), flush=True)
            if chaos_only:
                continue
            for tag, p, seeds in (("p01", 0.1, SEEDS_BAND), ("p00", 0.0, (10,)), ("p02", 0.2, SEEDS_P02)):
                t0 = time.time()
                runs = [train_run(mk, fwd, tgt, init, s, p) for s in seeds]
                out["%s.%s.%s.err_cm" % (side, stage, tag)] = np.asarray([r[0] for r in runs])
                out["%s.%s.%s.loss_curve" % (side, stage, tag)] = np.asarray([r[1] for r in runs])
                print("%-6s %-5s %s err_cm %s  (%.0f s)" % (side, stage, tag, np.round([r[0] for r in runs], 4), time.time() - t0), flush=True)
    np.savez_compressed(path, **out)
    print("wrote", path)


if __name__ == "__main__":
    main()
