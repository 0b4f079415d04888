"""Torch-only reproduction (no launch goes through libmmego_hip) of the graph-capture rule behind ops.capture /
ops.capture_can_fork: during torch.cuda.graph capture a stream may be forked from the capture's ORIGIN stream (modes C, E:
fine), while an event recorded on an already-forked stream and waited on by another non-origin stream (modes A, A_nok, B, D)
aborted the process on ROCm 7.2 / torch 2.10+rocm7.0 / MI355X.

  python scripts/repro_nested_capture_fork.py            -> runs the PASSING modes C and E
  python scripts/repro_nested_capture_fork.py A --allow-abort
       -> one failing mode, explicitly; installs a native-backtrace handler (scripts/abort_bt.c, built with gcc on the spot) so
          that the abort names its own faulting frame; run it under AMD_LOG_LEVEL=3 to get the last HIP API calls as well.
The kernels are torch fills: nothing of this repo runs between capture begin and end."""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
modes = args or ["C", "E"]
if any(m in ("A", "A_nok", "B", "D") for m in modes):
    if "--allow-abort" not in sys.argv:
        raise SystemExit("modes A, A_nok, B, D abort the process on the affected ROCm: pass --allow-abort to run one on purpose")
    so = "/tmp/libabort_bt.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-o", so, os.path.join(HERE, "abort_bt.c")], check=True)
    ctypes.CDLL(so).abort_bt_install()
    import faulthandler
    faulthandler.enable()

dev = torch.device("cuda:0")
bufs = [torch.zeros(1 << 16, device=dev) for _ in range(16)]


def k(i):
    bufs[i].fill_(1.0)


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def body(mode):
    cur = torch.cuda.current_stream()
    if mode == "A":          # main -> s1 -> s2 -> s1 -> main, nothing else
        k(0)
        s1.wait_stream(cur)
        with torch.cuda.stream(s1):
            k(1)
            s2.wait_stream(s1)
            with torch.cuda.stream(s2):
                k(2)
            s1.wait_stream(s2)
            k(3)
        cur.wait_stream(s1)
        k(4)
    elif mode == "A_nok":    # same, but no kernel on s1 before forking s2
        k(0)
        s1.wait_stream(cur)
        with torch.cuda.stream(s1):
            s2.wait_stream(s1)
            with torch.cuda.stream(s2):
                k(2)
            k(1)
            s1.wait_stream(s2)
        cur.wait_stream(s1)
        k(4)
    elif mode == "B":        # A + work on main in between
        k(0)
        s1.wait_stream(cur)
        with torch.cuda.stream(s1):
            k(1)
            s2.wait_stream(s1)
            with torch.cuda.stream(s2):
                k(2)
            k(5)
            s1.wait_stream(s2)
            k(3)
        k(6)
        cur.wait_stream(s1)
        k(4)
    elif mode == "C":        # one level but TWO forks from main, each with interleaved launches
        s1.wait_stream(cur); s2.wait_stream(cur)
        for j in range(3):
            k(0)
            with torch.cuda.stream(s1):
                k(1)
            with torch.cuda.stream(s2):
                k(2)
        cur.wait_stream(s1); cur.wait_stream(s2)
    elif mode == "D":        # fork s1; inside s1 repeated fork/join of s2 (3 times)
        s1.wait_stream(cur)
        with torch.cuda.stream(s1):
            for j in range(3):
                s2.wait_stream(s1)
                k(1)
                with torch.cuda.stream(s2):
                    k(2)
                s1.wait_stream(s2)
        cur.wait_stream(s1)
    elif mode == "E":        # one-level: repeated fork/join of s1 from main 3 times
        for j in range(3):
            s1.wait_stream(cur)
            k(0)
            with torch.cuda.stream(s1):
                k(1)
            cur.wait_stream(s1)
    else:
        raise SystemExit("unknown mode " + mode)


for mode in modes:
    main = torch.cuda.Stream()
    with torch.cuda.stream(main):
        body(mode)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        print(mode, "capture...", flush=True)
        with torch.cuda.graph(g, stream=main):
            body(mode)
        print(mode, "instantiated", flush=True)
        g.replay()
        torch.cuda.synchronize()
    print(mode, "ok", flush=True)
