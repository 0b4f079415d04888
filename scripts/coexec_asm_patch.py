"""r06: in-situ test of the hazard hypothesis for the co-residency finding -- the REAL head_fk_loss_kernel<1>, as compiled,
with single instructions of its assembly patched.

  python scripts/coexec_asm_patch.py build        (build container: csrc/geom.hip -> device assembly -> patched variants -> code objects
                                                   mmego_amd/lib/variants/hfk_<name>.hsaco; they travel with the snapshot)
  python scripts/coexec_asm_patch.py run [rounds] (GPU box: every code object through hipModuleLoad / hipModuleLaunchKernel beside the
                                                   16-unit bf16-MFMA step kernel of the product library, as scripts/coexec_variants.py)

Variants (only the body of head_fk_loss_kernel<1> is touched; `base` is the unpatched assembly and must reproduce the finding):
  vccpk    s_nop 1 in front of every VALU reader of VCC (v_cndmask_b32 / v_div_fmas_f32) whose VCC was written by a v_cmp / v_div_scale
           at most 3 instructions earlier with a PACKED-fp32 instruction in between (the sequences without an s_nop of the compiler's:
           it counts the packed instruction as one of the two wait states the gfx940 rule asks for)
  vccall   s_nop 1 in front of every VALU reader of VCC
  pkpad    s_nop 0 behind every packed-fp32 instruction
  pkpad3   s_nop 3 behind every packed-fp32 instruction
  pkin3    s_nop 3 behind the packed-fp32 instructions that sit BETWEEN a VALU write of VCC and the first VALU reader of that VCC
  pkout3   s_nop 3 behind all the others
  anyin3   s_nop 3 behind the NON-packed VALU instructions between a VALU write of VCC and its first reader (control for pkin3)
  depk_in  every packed-fp32 instruction BETWEEN a VALU write of VCC and the first VALU reader of that VCC replaced by its two (three)
           scalar equivalents (same operands, same IEEE results: the runner checks the idle-GPU outputs against `base` bit for bit)
  depk_out the same for the packed instructions OUTSIDE such windows;  depk_all: for all of them
  probe    depk_all + CAPTURES around the one packed instruction depk_all cannot take apart (its destination pair is its second source
           pair with the halves crossed): operands before, results after, in six extra VGPRs stored over dy columns 0-5 at the end of the
           kernel (`run` prints them for the rows whose column 41 differs, with the candidate expressions the wrong result equals)
  uncross  base with ONLY the in-place crossed packed instructions (destination pair = a source pair, op_sel crossing the halves)
           rewritten through two scratch VGPRs (v_mov copies of the overlapped pair first); every other packed instruction untouched
"""
import ctypes
import os
import re
import struct
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "mmego_amd", "lib", "variants")
KERNEL = "_Z19head_fk_loss_kernelILi1EEvPKfS1_ilPfS2_S1_S1_S2_PxiPyS1_PKiifS2_S2_PdPj"
LLVM = "/opt/rocm/lib/llvm/bin"
PK = re.compile(r"^\s*v_pk_(fma|mul|add)_f32|^\s*v_pk_mov_b32")
VCC_WRITE = re.compile(r"^\s*(v_cmp\w*\s+vcc,|v_div_scale_f32\s+v\d+,\s*vcc,)")
VCC_READ = re.compile(r"^\s*(v_cndmask_b32\w*\s+.*\bvcc\s*$|v_div_fmas_f32\b)")


def is_insn(line):
    t = line.strip()
    return bool(t) and not t.startswith((";", ".", "//")) and not t.endswith(":")


PKRE = re.compile(r"^\s*v_pk_(mul_f32|add_f32|fma_f32|mov_b32)\s+v\[(\d+):(\d+)\],\s*v\[(\d+):(\d+)\],\s*v\[(\d+):(\d+)\](?:,\s*v\[(\d+):(\d+)\])?(.*)$")


def depack(line):
    """A packed-fp32 instruction on VGPR pairs -> the equivalent scalar instructions (None: a form this does not handle)."""
    m = PKRE.match(line)
    if not m:
        return None
    op = m.group(1)
    d0, d1 = int(m.group(2)), int(m.group(3))
    srcs = [(int(m.group(4)), int(m.group(5))), (int(m.group(6)), int(m.group(7)))]
    if m.group(8) is not None:
        srcs.append((int(m.group(8)), int(m.group(9))))
    n = len(srcs)
    mods = {"op_sel": [0] * n, "op_sel_hi": [1] * n, "neg_lo": [0] * n, "neg_hi": [0] * n}
    for name, vals in re.findall(r"(op_sel_hi|op_sel|neg_lo|neg_hi):\[([0-9,]+)\]", m.group(10)):
        v = [int(x) for x in vals.split(",")]
        mods[name] = v + mods[name][len(v):]
    if op == "mov_b32":
        lo_src, hi_src = [srcs[0][mods["op_sel"][0]]], [srcs[1][mods["op_sel"][1]]]
        lo = "\tv_mov_b32_e32 v%d, v%d\n" % (d0, lo_src[0])
        hi = "\tv_mov_b32_e32 v%d, v%d\n" % (d1, hi_src[0])
    else:
        lo_src = [srcs[i][mods["op_sel"][i]] for i in range(n)]
        hi_src = [srcs[i][mods["op_sel_hi"][i]] for i in range(n)]
        name = {"mul_f32": "v_mul_f32_e64", "add_f32": "v_add_f32_e64", "fma_f32": "v_fma_f32"}[op]
        fmt = lambda d, ss, neg: "\t%s v%d, %s\n" % (name, d, ", ".join(("-" if neg[i] else "") + "v%d" % ss[i] for i in range(n)))
        lo, hi = fmt(d0, lo_src, mods["neg_lo"]), fmt(d1, hi_src, mods["neg_hi"])
    if d0 in hi_src:
        if d1 in lo_src:
            return None
        return [hi, lo]
    return [lo, hi]


def crossed_in_place(line):
    """-> index of the source pair that IS the destination pair with a half crossed (the low result's register is read by the high
    half, or the high result's by the low half), or None."""
    m = PKRE.match(line)
    if not m or m.group(1) == "mov_b32":
        return None
    d = (int(m.group(2)), int(m.group(3)))
    srcs = [(int(m.group(4)), int(m.group(5))), (int(m.group(6)), int(m.group(7)))]
    if m.group(8) is not None:
        srcs.append((int(m.group(8)), int(m.group(9))))
    n = len(srcs)
    mods = {"op_sel": [0] * n, "op_sel_hi": [1] * n}
    for name, vals in re.findall(r"(op_sel_hi|op_sel):\[([0-9,]+)\]", m.group(10)):
        v = [int(x) for x in vals.split(",")]
        mods[name] = v + mods[name][len(v):]
    for i, sp in enumerate(srcs):
        if sp == d and (mods["op_sel"][i] == 1 or mods["op_sel_hi"][i] == 0):
            return i
    return None


def uncross(lines):
    out, n = [], 0
    for line in lines:
        i = crossed_in_place(line) if is_insn(line) else None
        if i is None:
            out.append(line)
            continue
        m = PKRE.match(line)
        d0, d1 = int(m.group(2)), int(m.group(3))
        out += ["\tv_mov_b32_e32 v214, v%d\n" % d0, "\tv_mov_b32_e32 v215, v%d\n" % d1]
        head, _, rest = line.partition("v[%d:%d]," % (d0, d1))             # keep the destination, replace the matching source pair(s)
        out.append(head + "v[%d:%d]," % (d0, d1) + rest.replace("v[%d:%d]" % (d0, d1), "v[214:215]"))
        n += 1
    return out, n


def patch(lines, kind):
    out, hist, n = [], [], 0                        # hist: the instructions since the last VCC write (None before the first)
    for line in lines:
        if not is_insn(line):
            if line.strip().endswith(":"):
                hist = None if hist is None else hist + ["<label>"]
            out.append(line)
            continue
        if kind in ("vccpk", "vccall") and VCC_READ.match(line) and hist is not None:
            near = len(hist) <= 3 and not any(h.strip().startswith("s_nop") for h in hist)
            if kind == "vccall" or (near and any(PK.match(h) for h in hist)):
                out.append("\ts_nop 1\n")
                n += 1
        if kind in ("depk_in", "depk_out", "depk_all") and PK.match(line):
            inside = hist is not None and not any(VCC_READ.match(h) for h in hist if h != "<label>")
            if kind == "depk_all" or inside == (kind == "depk_in"):
                rep = depack(line)
                if rep is not None:
                    out.extend(rep)
                    n += 1
                    if hist is not None:
                        hist.extend(rep)
                    continue
        out.append(line)
        if kind == "pkpad" and PK.match(line):
            out.append("\ts_nop 0\n")
            n += 1
        if kind == "pkpad3" and PK.match(line):
            out.append("\ts_nop 3\n")
            n += 1
        if kind in ("pkin3", "pkout3", "anyin3"):
            inside = hist is not None and not any(VCC_READ.match(h) for h in hist if h != "<label>")
            want = (kind == "pkin3" and PK.match(line) and inside) or (kind == "pkout3" and PK.match(line) and not inside) or \
                   (kind == "anyin3" and inside and line.strip().startswith("v_") and not PK.match(line) and not VCC_READ.match(line) and not VCC_WRITE.match(line))
            if want:
                out.append("\ts_nop 3\n")
                n += 1
        if VCC_WRITE.match(line):
            hist = []
        elif hist is not None:
            hist.append(line)
    return out, n


CROSSED = "v_pk_add_f32 v[12:13], v[22:23], v[12:13] op_sel:[0,1] op_sel_hi:[1,0]"


def add_probe(body):
    """-> body with the captures of variant `probe`: the operands (v12, v13, v22, v23) in front of and the results (v12, v13) behind the
    one packed instruction that `depk_all` cannot take apart -- the IN-PLACE, CROSSED add  v12' = v22 + v13,  v13' = v23 + v12  that sits in
    the dataflow of dy columns 40 / 41 -- stored over dy columns 0-5 at the end of the kernel."""
    out, state = [], 0
    for line in body:
        t = line.strip()
        if state == 0 and t == CROSSED:
            out += ["\tv_mov_b32_e32 v214, v12\n", "\tv_mov_b32_e32 v215, v13\n", "\tv_mov_b32_e32 v216, v22\n", "\tv_mov_b32_e32 v217, v23\n"]
            out.append(line)
            out += ["\tv_mov_b32_e32 v218, v12\n", "\tv_mov_b32_e32 v219, v13\n"]
            state = 1
            continue
        out.append(line)
        if state == 1 and t == "global_store_dwordx4 v[34:35], v[6:9], off offset:144":
            out += ["\ts_nop 1\n", "\tglobal_store_dwordx4 v[34:35], v[214:217], off\n", "\tglobal_store_dwordx2 v[34:35], v[218:219], off offset:16\n"]
            state = 2
    assert state == 2, state
    return out


def build():
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(OUT, "geom_dev.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=on", "-S", "--cuda-device-only",
                    os.path.join(ROOT, "mmego_amd", "csrc", "geom.hip"), "-o", src], check=True, stderr=subprocess.DEVNULL)
    lines = open(src).readlines()
    a = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    b = next(i for i in range(a, len(lines)) if lines[i].strip().startswith(".amdhsa_kernel " + KERNEL))
    for kind in ("base", "depk_all", "probe", "uncross"):
        if kind == "uncross":
            body, n = uncross(lines[a:b])
            for l in lines[a:b]:
                if is_insn(l) and crossed_in_place(l) is not None:
                    print("   crossed in place:", l.strip())
        else:
            body, n = (lines[a:b], 0) if kind == "base" else patch(lines[a:b], "depk_all" if kind == "probe" else kind)
        tail = lines[b:]
        if kind == "probe":
            body = add_probe(body)
        if kind in ("probe", "uncross"):
            e = next(i for i, l in enumerate(tail) if ".end_amdhsa_kernel" in l)
            tail = [l.replace(".amdhsa_next_free_vgpr 214", ".amdhsa_next_free_vgpr 220").replace(".amdhsa_accum_offset 216", ".amdhsa_accum_offset 220")
                    if i < e else l for i, l in enumerate(tail)]
            assert any(".amdhsa_next_free_vgpr 220" in l for l in tail)
        s = os.path.join(OUT, "hfk_%s.s" % kind)
        open(s, "w").writelines(lines[:a] + body + tail)
        o = s[:-2] + ".o"
        subprocess.run([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o], check=True)
        subprocess.run([LLVM + "/ld.lld", "-shared", o, "-o", s[:-2] + ".hsaco"], check=True)
        os.remove(o)
        print("%-8s %4d patches -> %s" % (kind, n, s[:-2] + ".hsaco"))
    os.remove(src)


def run(rounds):
    import glob
    import torch
    sys.path.insert(0, ROOT)
    from mmego_amd import blocks, hip
    dev = torch.device("cuda:0")
    hip.lib()
    rt = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    Bn, S, H = 512, 20, 512
    lstm = blocks.LstmParams(H, H, 2, dropout=0.0, bidirectional=True).to(dev)
    xs = torch.randn(Bn * S, H, device=dev).relu_()
    sB = torch.cuda.Stream()
    nrb, S2 = Bn // 32, 2 * H // 16
    d = dict(W=blocks.lstm_split3_weights(lstm, 16), x=blocks.split3_cvt(xs, tm=(Bn, S, Bn)), xpf=torch.empty(S * Bn * 8 * H, device=dev),
             O=[blocks.split3_cvt(torch.zeros(S * Bn, 2 * H, device=dev)) for _ in range(2)], out=torch.empty(Bn * S, 2 * H, device=dev),
             c=torch.zeros(2, Bn, H, device=dev))

    def step16_stack():
        cur, K = d["x"], H
        for layer in range(2):
            wih, bias, whh0, whh1 = d["W"][layer]
            hip.call("split3_gemm", cur, wih, d["xpf"], None, 0, bias, S * nrb, 8 * H // 32, K, 0, 6, 0)
            o_p, out_p = d["O"][layer].data_ptr(), d["out"].data_ptr()
            win = lambda tt, dd: o_p + 2 * ((tt * nrb * S2 + dd * (H // 16)) * 3 * 512)
            ho = lambda tt, dd: out_p + 4 * (tt * 2 * H + dd * H) if layer == 1 else None
            for s_ in range(S):
                t0, t1 = s_, S - 1 - s_
                hip.call("split3_step16", 2, Bn, H, int(s_ == 0), win(t0 - 1, 0) if s_ else None, win(t1 + 1, 1) if s_ else None, S2 * 3,
                         whh0, whh1, d["xpf"], t0 * nrb, t1 * nrb, ho(t0, 0), ho(t1, 1), S * 2 * H, win(t0, 0), win(t1, 1), S2 * 3,
                         d["c"][0], d["c"][1], 6, 0)
            cur, K = d["O"][layer], 2 * H

    with torch.cuda.stream(sB), torch.no_grad():
        step16_stack()
    torch.cuda.synchronize()
    B, F = 64, 512
    g = torch.Generator().manual_seed(3)
    R = torch.linalg.qr(torch.randn(F, 3, 3, generator=g))[0].contiguous().to(dev)
    t = (torch.randn(F, 3, generator=g) * 0.1).to(dev)
    y = torch.randn(F, 42, generator=g).to(dev)
    body = (torch.randn(B, 20, 3, generator=g) * 0.2).to(dev)
    target = torch.randn(F, 21, 3, generator=g).to(dev)
    jmap = torch.tensor([12, 13, 14, 15, 16, 17, 18, 19], dtype=torch.int32, device=dev)
    q = torch.zeros(F, 6, 3, 3, device=dev)
    jh, l = torch.zeros(F, 8, 3, device=dev), torch.zeros(F, 8, 3, device=dev)
    loss2, dy, scr = torch.zeros(2, device=dev), torch.zeros(F, 42, device=dev), torch.zeros(17, dtype=torch.float64, device=dev)
    outs = (q, jh, l, dy, loss2)
    # the kernel's explicit arguments (offsets from the code object's metadata: 152 bytes)
    P = lambda x: x.data_ptr() if x is not None else 0
    nb = (F + 63) // 64
    karg = struct.pack("<QQi4xqQQQQQQi4xQQQifQQQQ", P(y), P(body), B, F, P(q), P(jh), P(R), P(t), P(l), 0, 0, 0, P(target), P(jmap), 21, 1.0,
                       P(loss2), P(dy), P(scr), P(scr) + 8 * 2 * nb)
    assert len(karg) == 152
    kbuf = ctypes.create_string_buffer(karg, len(karg))
    ksize = ctypes.c_size_t(len(karg))
    extra = (ctypes.c_void_p * 5)(1, ctypes.cast(kbuf, ctypes.c_void_p).value, 2, ctypes.cast(ctypes.pointer(ksize), ctypes.c_void_p).value, 3)
    rt.hipModuleLaunchKernel.argtypes = [ctypes.c_void_p] + [ctypes.c_uint] * 7 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    base_ref = None
    for path in sorted(glob.glob(os.path.join(OUT, "hfk_*.hsaco")), key=lambda q: (not q.endswith("hfk_base.hsaco"), q)):
        name = os.path.basename(path)[4:-6]
        mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
        assert rt.hipModuleLoad(ctypes.byref(mod), path.encode()) == 0
        assert rt.hipModuleGetFunction(ctypes.byref(fn), mod, KERNEL.encode()) == 0

        def victim():
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(20):
                rc = rt.hipModuleLaunchKernel(fn, nb, 1, 1, 64, 1, 1, 0, st, None, extra)
                assert rc == 0, rc

        for o in outs:
            o.zero_()
        victim()
        torch.cuda.synchronize()
        ref = [o.clone() for o in outs]
        assert ref[3].abs().max().item() > 0
        if base_ref is None:
            base_ref = ref
        same = all(torch.equal(a, b) for a, b in zip(ref, base_ref))
        bad, badcols, shown = 0, {}, 0
        for it in range(rounds):
            with torch.cuda.stream(sB), torch.no_grad():
                step16_stack()
            victim()
            torch.cuda.synchronize()
            if any(not torch.equal(a, b) for a, b in zip(outs, ref)):
                bad += 1
                for c_ in set((dy != ref[3]).nonzero()[:, 1].tolist()):
                    badcols[c_] = badcols.get(c_, 0) + 1
                if name == "probe" and shown < 8:
                    rows = sorted(set((dy[:, 41] != ref[3][:, 41]).nonzero().view(-1).tolist()))
                    for r_ in rows[:2] + rows[-1:]:
                        v12, v13, v22, v23, o12, o13 = dy[r_, :6].cpu().tolist()
                        f32 = lambda x: torch.tensor(x, dtype=torch.float32)
                        want12, want13 = (f32(v22) + f32(v13)).item(), (f32(v23) + f32(v12)).item()
                        cands = {"v22 + v13 (right)": want12, "v22 + v12": (f32(v22) + f32(v12)).item(), "v22 + v13' (= v22 + v23 + v12)": (f32(v22) + f32(want13)).item(),
                                 "v22": v22, "v13": v13, "v12 (unchanged)": v12, "v23 + v12": want13, "v23 + v13": (f32(v23) + f32(v13)).item()}
                        hit = [k for k, v in cands.items() if v == o12]
                        print("   row %3d (lane %2d): in v12 %+.7f v13 %+.7f v22 %+.7f v23 %+.7f -> out v12' %+.7f (right: %+.7f; equals: %s)   v13' %+.7f (right: %+.7f)" % (
                            r_, r_ % 64, v12, v13, v22, v23, o12, want12, hit, o13, want13), flush=True)
                    shown += 1
        print("code object %-8s (idle-GPU outputs %s base's): %3d of %d rounds differ; rounds per differing dy column %s" % (
            name, "bit-equal to" if same else "DIFFERENT from", bad, rounds, dict(sorted(badcols.items()))), flush=True)
        rt.hipModuleUnload(mod)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 600)
