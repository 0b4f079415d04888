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
# No packed-fp32 instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_pk_mov_b32) in any kernel: the target feature is switched
# off for the device compilation.  r06 finding: such an instruction whose op_sel takes the HIGH register of its
# second source pair for the LOW result can receive 0.0 for that operand in lanes 48-63 while s3_gemm_kernel's workgroups are resident
# on the same CU (standalone: scripts/coexec_pk_probe.hip, 45-92 of 1000 rounds; the whole step: silently wrong gradients in 22-59 of 60
# engines with the instructions, 0 of 120 without).  Which packed instruction gets that op_sel is the register allocator's choice, so the
# class goes; tests/test_host_cpu.py holds the built library to zero of them.  No measurable cost on the figures bench.py reports (U+L
# step 4.98-4.99 ms with, 5.00 ms without; config 5, wlocal and stage-1 within box-to-box noise); per kernel, the VALU-heavy bf16 ones
# of config 5 pay up to 9 % (gcn_mix_eval_bf16 208 -> 226 us), and ONE kernel had started to spill registers without the packed forms
# (attn_pool_frag_bf16: 281 -> 387 us) until its loop stopped keeping 64 unpacked values alive (282 us again).
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=" + ARCH, "-fPIC", "-ffp-contract=on", "-Wall", "-Wno-unused-function"] + NO_PACKED_FP32 + os.environ.get("MMEGO_EXTRA_HIPCC_FLAGS", "").split()
# Per-file additions.  front_bf16.hip: its 16 x 16 MFMA chains hand every accumulator straight to VALU code (ReLU, bf16 rounding, the next
# MFMA's operand); with the accumulators in AGPRs -- the compiler's default choice -- each of the 52 values per slab costs a
# v_accvgpr_read_b32 first (r06: 342 -> 290 instructions per slab and wave).
# -fno-honor-nans there: relu(x) is one v_max_f32 instead of two (the first canonicalises a possible signalling NaN); the file holds one
# eval-mode kernel whose inputs are finite activations.
_VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
# (front.hip, the fp32 form of the same kernel: accumulators in VGPRs too, but NaNs stay honoured there -- a NaN in the input of the
#  default path has to come out as a NaN, as it does from the reference)
FILE_FLAGS = {"front_bf16.hip": _VGPR_FORM + ["-fno-honor-nans"], "front.hip": _VGPR_FORM}
# (the host pass of the same command line does not know the amdgcn feature and says so once per file)
_HOST_NOISE = ("'-packed-fp32-ops' is not a recognized feature for this target (ignoring feature)", "1 warning generated when compiling for host.")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True, variant=None, extra_flags=()):
    """variant (measurement builds only, scripts/): a second library lib/variants/libmmego_hip_<variant>.so with `extra_flags` appended,
    objects in build_<variant>/; the product library is variant None."""
    lib = LIB if variant is None else os.path.join(LIBDIR, "variants", "libmmego_hip_%s.so" % variant)
    if variant is None and not force and not stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    objdir = os.path.join(HERE, "build" if variant is None else "build_" + variant)
    os.makedirs(objdir, exist_ok=True)

    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    newest_h = max([os.path.getmtime(h) for h in headers] + [os.path.getmtime(os.path.abspath(__file__))])

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), newest_h):
            return obj                  # (object newer than its source, every header and this recipe: keep it)
        cmd = [hipcc] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + list(extra_flags) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr))
        err = "\n".join(l for l in r.stderr.splitlines() if l.strip() and not any(n in l for n in _HOST_NOISE))
        if verbose and err.strip():
            sys.stderr.write(err + "\n")
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, sources()))
    r = subprocess.run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    if verbose:
        print("built", lib)
    return lib


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
