"""Count the signs of serialized memory round trips in a kernel's gfx950 assembly: `s_waitcnt vmcnt(0)`, `s_cbranch_execz`
around loads (predicated loads compile to branch + load + wait), scratch use.
usage: python scripts/isa_scan.py mmego_amd/csrc/<file>.hip [more.hip ...]   (cross-compiles with hipcc -S; no GPU needed)"""
import os
import re
import subprocess
import sys
import tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for src in sys.argv[1:]:
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", "-w",
                        "-I" + os.path.join(root, "mmego_amd/csrc"), "-I" + os.path.join(root, "include"), src, "-o", out], check=True)
        txt = open(out).read()
    print(src)
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end", txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        first_store = body.find("global_store")
        after = body[first_store:] if first_store >= 0 else ""
        print("  %-56s lines %5d  vmcnt(0) %3d (%3d behind the first store)  execz %3d  loads %3d  stores %3d  scratch %3d" % (
            dem[:56], body.count("\n"), len(re.findall(r"vmcnt\(0\)", body)), len(re.findall(r"vmcnt\(0\)", after)),
            body.count("s_cbranch_execz"), len(re.findall(r"\b(global_load|buffer_load|flat_load)", body)),
            len(re.findall(r"\b(global_store|buffer_store|flat_store)", body)), len(re.findall(r"scratch_", body))))
