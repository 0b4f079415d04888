"""Build libmmego_hip.so (gfx950 only) in-tree with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so is git-ignored but
travels to the GPU box with the source snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmmego_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "--offload-arch=" + ARCH, "-fPIC", "-ffp-contract=on", "-Wall", "-Wno-unused-function"] + os.environ.get("MMEGO_EXTRA_HIPCC_FLAGS", "").split()


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True):
    if not force and not stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)

    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    newest_h = max([os.path.getmtime(h) for h in headers] + [os.path.getmtime(os.path.abspath(__file__))])

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), newest_h):
            return obj                  # (object newer than its source, every header and this recipe: keep it)
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, sources()))
    r = subprocess.run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
