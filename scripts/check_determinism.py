"""Run-to-run determinism of the HIP training path: three Adam steps of UpperNetwlocal / UpperNet / LowerNet on the g6 fixture,
print a checksum of the parameters after each step (bit patterns).  Run it several times: the lines must be identical."""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import nets  # noqa: E402
from mmego_amd.nets_local import UpperNetwlocal  # noqa: E402
from mmego_amd.params import FusedAdam  # noqa: E402

dev = torch.device("cuda:0")
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g6_train.npz"))
T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32).to(dev)
x0, body, R, t, target = [T(g[k]) for k in ("x", "body", "R", "t", "target")]
UM = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 16, 20]
for name, ctor in (("wlocal", UpperNetwlocal), ("upper", nets.UpperNet)):
    torch.manual_seed(602)
    m = ctor().to(dev).train()
    m.lstm_dropout = 0.0
    opt = FusedAdam(m.flat(), lr=3e-5)
    h0 = torch.zeros(6, 4, 64, device=dev)
    sums = []
    for step in range(3):
        if name == "wlocal":
            out = m(x0.clone(), h0, h0.clone(), h0.clone(), h0.clone(), body, R, t)[0]
        else:
            out = m(x0.clone(), h0, h0.clone(), body, R, t)[0]
        loss = (out - target[:, :, UM]).abs().sum()
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        sums.append("%s:%s" % (hashlib.md5(out.detach().cpu().numpy().tobytes()).hexdigest()[:8],
                               hashlib.md5(m.flat().flat_p.cpu().numpy().tobytes()).hexdigest()[:8]))
    print(name, " ".join(sums))
