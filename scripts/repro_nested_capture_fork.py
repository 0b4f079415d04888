"""Minimal reproduction of the ROCm graph-capture defect that shapes the stream topology of this repo: during torch.cuda.graph
capture, a stream may be forked from the capture's ORIGIN stream (modes C, E: fine), but forking a further stream from an
already-forked stream (modes A, A_nok, B, D: an event recorded on a fork, waited on by a non-origin stream) dumps core
(ROCm 7.2, torch 2.10+rocm7.0, MI355X).  usage: python scripts/repro_nested_capture_fork.py {A,A_nok,B,C,D,E}"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import hip, ops
dev = torch.device("cuda:0")
mode = sys.argv[1]
bufs = [torch.zeros(1 << 16, device=dev) for _ in range(16)]
def k(i):
    ops.fill(bufs[i], 1.0)
main = torch.cuda.Stream()
s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
def body():
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
            with torch.cuda.stream(s1): k(1)
            with torch.cuda.stream(s2): k(2)
        cur.wait_stream(s1); cur.wait_stream(s2)
    elif mode == "D":        # fork s1; inside s1 repeated fork/join of s2 (3 times)
        s1.wait_stream(cur)
        with torch.cuda.stream(s1):
            for j in range(3):
                s2.wait_stream(s1)
                k(1)
                with torch.cuda.stream(s2): k(2)
                s1.wait_stream(s2)
        cur.wait_stream(s1)
    elif mode == "E":        # one-level: repeated fork/join of s1 from main 3 times
        for j in range(3):
            s1.wait_stream(cur)
            k(0)
            with torch.cuda.stream(s1): k(1)
            cur.wait_stream(s1)
with torch.cuda.stream(main):
    body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    print(mode, "capture...", flush=True)
    with torch.cuda.graph(g, stream=main):
        body()
    print(mode, "instantiated", flush=True)
    g.replay()
    torch.cuda.synchronize()
print(mode, "ok", flush=True)
