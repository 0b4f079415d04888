"""Watchdog for the intermittent stall of the CPU suite.  Enabled by MMEGO_STALL_PROBE=<seconds>[:<logfile>]
(tests/conftest.py arms it around every test): when a test runs longer than that, every thread of the process gets a signal whose
handler (scripts/stall_bt.c) prints its NATIVE backtrace, three samples two seconds apart, together with each task's state and CPU time
from /proc -- what the OpenMP / MKL workers were doing, which the Python-level dumps of pytest-timeout / faulthandler cannot show."""
import ctypes
import faulthandler
import os
import signal
import subprocess
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None
_fd = 2


def _load(logfile):
    global _lib, _fd
    if _lib is None:
        so = "/tmp/libstall_bt_%d.so" % os.getuid()
        src = os.path.join(ROOT, "scripts", "stall_bt.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-o", so, src], check=True)
        if logfile:
            _fd = os.open(logfile, os.O_WRONLY | os.O_CREAT | os.O_APPEND, 0o644)
        _lib = ctypes.CDLL(so)
        _lib.stall_bt_kick.argtypes = [ctypes.c_int, ctypes.c_long]
        _lib.stall_bt_install(int(signal.SIGUSR2), _fd)
    return _lib


def _tasks():
    out = []
    for tid in sorted(int(t) for t in os.listdir("/proc/self/task")):
        try:
            f = open("/proc/self/task/%d/stat" % tid).read()
            rest = f[f.rindex(")") + 2:].split()
            out.append((tid, rest[0], int(rest[11]), int(rest[12])))          # state, utime, stime (clock ticks)
        except OSError:
            pass
    return out


class Watchdog:
    def __init__(self, seconds, logfile, label):
        self.seconds, self.label, self.lib = seconds, label, _load(logfile)
        self.done = threading.Event()
        self.fired = False
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def _say(self, s):
        os.write(_fd, s.encode())

    def _run(self):
        if self.done.wait(self.seconds):
            return
        self.fired = True
        me = threading.get_native_id()
        self._say("\n===== [stall_probe] %s still running after %.0f s (pid %d) =====\n" % (self.label, self.seconds, os.getpid()))
        try:
            import torch
            self._say("torch threads %d, interop %d, flush_denormal probe: %s\n" % (torch.get_num_threads(), torch.get_num_interop_threads(), os.environ.get("OMP_NUM_THREADS")))
        except Exception as e:  # noqa: BLE001
            self._say("(torch query failed: %r)\n" % (e,))
        for sample in range(3):
            self._say("\n--- sample %d: tid state utime stime (ticks)\n" % sample)
            for tsk in _tasks():
                self._say("%d %s %d %d\n" % tsk)
            with open(_fd, "a", closefd=False) as fh:
                faulthandler.dump_traceback(file=fh, all_threads=True)
            for tid, *_ in _tasks():
                if tid != me:
                    self.lib.stall_bt_kick(int(signal.SIGUSR2), tid)
                    time.sleep(0.05)
            if self.done.wait(2.0):
                self._say("--- (the test finished while sampling)\n")
                return

    def stop(self):
        self.done.set()


def arm(label):
    spec = os.environ.get("MMEGO_STALL_PROBE")
    if not spec:
        return None
    secs, _, logfile = spec.partition(":")
    return Watchdog(float(secs), logfile or None, label)
