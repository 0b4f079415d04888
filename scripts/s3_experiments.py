"""Timing by elimination for the split3 kernels (mmego_amd/csrc/split3.hip): builds variant libraries with -DS3_EXP=<mask> (compile-time
masks documented at the top of split3.hip; results of a variant are WRONG by design) and times the projection GEMM and the recurrent
step of each.  `python scripts/s3_experiments.py build` cross-compiles the variants (no GPU needed) into scripts/exp/; `run` times
them on the GPU."""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EXP = os.path.join(HERE, "exp")
MASKS = (0, 1, 2, 3, 4, 8, 9, 11)


def build():
    os.makedirs(EXP, exist_ok=True)
    for m in MASKS:
        out = os.path.join(EXP, "libs3_%d.so" % m)
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=on", "-shared", "-DS3_EXP=%d" % m,
               "-I", os.path.join(ROOT, "mmego_amd", "csrc"), os.path.join(ROOT, "mmego_amd", "csrc", "split3.hip"), "-o", out]
        subprocess.run(cmd, check=True)
        print("built", out)


def run():
    import torch
    sys.path.insert(0, ROOT)
    from mmego_amd import blocks, hip, ops
    dev = torch.device("cuda:0")
    hip.lib()
    Bn, S, H = 512, 20, 512
    M, N = Bn * S, 8 * H
    torch.manual_seed(0)
    lstm = blocks.LstmParams(H, H, 2, dropout=0.0, bidirectional=True).to(dev)
    W3 = blocks.lstm_split3_weights(lstm)
    ar = ops.Arena(dev)
    nrb, S2 = Bn // 32, 2 * H // 16
    xpf = torch.randn(S * Bn * 8 * H, device=dev)
    O = blocks.split3_cvt(torch.randn(S * Bn, 2 * H, device=dev).tanh_())
    c = torch.zeros(2, Bn, H, device=dev)
    o_p = O.data_ptr()
    win = lambda t, d: o_p + 2 * ((t * nrb * S2 + d * (H // 16)) * 3 * 512)
    st = lambda: torch.cuda.current_stream().cuda_stream
    P = lambda t: ctypes.c_void_p(t.data_ptr() if isinstance(t, torch.Tensor) else (t or 0))

    def timeit(fn, n):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (5 * n) * 1e3

    gemm_ops = {}
    for K in (512, 1024):
        A = torch.randn(M, K, device=dev).relu_()
        W = torch.randn(N, K, device=dev) * 0.04
        gemm_ops[K] = (blocks.split3_cvt(A), blocks.split3_cvt(W), torch.empty(M * N, device=dev), torch.randn(N, device=dev))
    for m in MASKS:
        lib = ctypes.CDLL(os.path.join(EXP, "libs3_%d.so" % m))
        lib.mmego_split3_step.restype = ctypes.c_int
        lib.mmego_split3_gemm.restype = ctypes.c_int

        def steps():
            for s in range(S):
                t0, t1 = s, S - 1 - s
                rc = lib.mmego_split3_step(ctypes.c_void_p(st()), 2, Bn, H, int(s == 0), P(win(t0 - 1, 0) if s else 0), P(win(t1 + 1, 1) if s else 0),
                                           ctypes.c_long(S2 * 3), P(W3[0][2]), P(W3[0][3]), P(xpf), ctypes.c_long(t0 * nrb), ctypes.c_long(t1 * nrb),
                                           P(0), P(0), ctypes.c_long(0), P(win(t0, 0)), P(win(t1, 1)), ctypes.c_long(S2 * 3), P(c[0]), P(c[1]), 6, 0)
                assert rc == 0, rc
        us_step = timeit(steps, 2) / S
        line = "S3_EXP=%2d  step %6.2f us per timestep" % (m, us_step)
        for K in (512, 1024):
            Ap, Wp, Cf, bias = gemm_ops[K]
            for wm in (2, 4):
                def gemm():
                    rc = lib.mmego_split3_gemm(ctypes.c_void_p(st()), P(Ap), P(Wp), P(Cf), P(0), ctypes.c_long(0), P(bias), M // 32, N // 32, K, 0, 6, wm)
                    assert rc == 0, rc
                line += " | gemm K=%d wm=%d %6.1f us" % (K, wm, timeit(gemm, 10))
        print(line, flush=True)


if __name__ == "__main__":
    (build if sys.argv[1:] == ["build"] else run)()
