"""The descriptor structs of include/mmego_hip.h against their ctypes mirrors in mmego_amd/hip.py: the header is compiled with gcc (it is
plain C) into a program that prints sizeof and every field's offset; the Python classes must agree field by field.  (The kernels'
host-side mirrors inside the .hip files are held to the header by static_asserts or by sharing its definition.)"""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mmego_hip.h")
PAIRS = {"MmegoGemmDesc": "GemmDesc", "MmegoBnRef": "BnRef", "MmegoGcnFront": "GcnFront", "MmegoPack": "Pack", "MmegoDwRed": "DwRed",
         "MmegoSlab": "Slab", "MmegoLstm64Fwd": "Lstm64Fwd", "MmegoLstm64Bwd": "Lstm64Bwd"}


def _c_fields(text, name):
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", " ", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        # "const float* a", "int B, T", "const float* xproj[2]", "MmegoBnRef bn1": the first declarator is the last word of the first
        # comma segment, the others are the remaining segments
        segs = decl.split(",")
        for d in [segs[0].replace("*", " ").split()[-1]] + segs[1:]:
            fields.append(re.sub(r"[\*\s]|\[.*\]", "", d))
    return fields


def test_every_descriptor_struct_matches_its_ctypes_mirror(tmp_path):
    from mmego_amd import hip
    text = open(HEADER).read()
    declared = set(re.findall(r"typedef struct (\w+) \{", text))
    assert declared == set(PAIRS), ("a struct of the header has no entry in this test (or the other way round)", declared ^ set(PAIRS))
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "%s"' % HEADER, "int main(void) {"]
    for cname in PAIRS:
        lines.append('  printf("%s sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for f in _c_fields(text, cname):
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, f, cname, f))
    lines += ["  return 0;", "}"]
    src, exe = tmp_path / "abi.c", tmp_path / "abi"
    src.write_text("\n".join(lines))
    subprocess.run(["gcc", "-std=c11", "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    got = {}
    for ln in out.splitlines():
        s, f, v = ln.split()
        got.setdefault(s, {})[f] = int(v)
    for cname, pyname in PAIRS.items():
        cls = getattr(hip, pyname)
        assert ctypes.sizeof(cls) == got[cname]["sizeof"], (cname, ctypes.sizeof(cls), got[cname]["sizeof"])
        py = {n: getattr(cls, n).offset for n, _ in cls._fields_}
        c = {k: v for k, v in got[cname].items() if k != "sizeof"}
        assert list(py) == list(c), (cname, "field names / order", list(py), list(c))
        assert py == c, (cname, py, c)
