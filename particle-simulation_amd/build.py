"""Build driver: compiles the HIP library and the C host tools in-tree.

    python particle-simulation_amd/build.py [--force] [--sanitize]

Outputs (all git-ignored, all shipped to the GPU box with the tree):
    particle-simulation_amd/libpaircount.so   HIP kernels + C ABI (include/paircount.h)
    particle-simulation_amd/libxyzhost.so     host tokenizer + synthetic-input generator
    particle-simulation_amd/main              C host program, the reference's CLI
After compiling, the device ISA is scanned: any fused binary32 multiply-add in
the pair-test or key kernels would change results, so the build fails on one.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
ARCH = "gfx950"

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CC = os.environ.get("CC", "gcc")

# -fno-slp-vectorize: the dense pair test's nine plain instructions per 64 tests must not be re-packed into
# two-wide instructions with register shuffles (a packed binary32 instruction takes two issue slots of a plain
# one, 4.2 against 2.35 cycles per wave and SIMD: profiles/r3c_valu_issue_rate.txt, tools/micro/valu_rate.hip)
HIP_FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-Wall"] + os.environ.get("PC_EXTRA_FLAGS", "").split()
# x86-64-v3 (AVX2, BMI2, LZCNT): what the tokenizer's bit tricks compile to single instructions on; the
# host CPUs of MI355X machines all have it.  (The reference builds with -O3 -march=native.)
C_FLAGS = ["-O3", "-march=x86-64-v3", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-pthread",
           "-Wall", "-Wextra"]

# any encoding suffix (_e32, _e64, _dpp, _sdwa) counts
FUSED_F32 = re.compile(r"\bv_(pk_)?(fma|fmac|mad|mac|fma_mix|fmamk|fmaak)_(legacy_)?f32")
# kernels whose only fused operations are the compiler's expansion of 64-bit integer division (exact by construction)
FUSED_F32_EXEMPT = ("pc_gen_points", "pc_t2_parse_records")     # (neither holds any binary32 arithmetic of its own)
FUNC_LABEL = re.compile(r"^(_Z\w+|\w+):\s*(;.*)?$")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build step failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def _sources(d, exts):
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(exts)]


def build_hip(force=False):
    out = os.path.join(HERE, "libpaircount.so")
    srcs = _sources(CSRC, (".hip", ".h")) + [os.path.join(ROOT, "include", "paircount.h")]
    if not (force or _newer(out, srcs)):
        return out
    main_src = os.path.join(CSRC, "paircount.hip")
    asm = os.path.join(HERE, "build", "paircount_gfx950.s")
    os.makedirs(os.path.dirname(asm), exist_ok=True)
    _run([HIPCC] + HIP_FLAGS + ["-S", "--cuda-device-only", "-Wno-unused-command-line-argument", "-o", asm, main_src])
    bad, func = [], ""
    with open(asm) as f:
        for ln in f:
            m = FUNC_LABEL.match(ln)
            if m and not ln.startswith(".L"):
                func = m.group(1)
            if FUSED_F32.search(ln) and not any(e in func for e in FUSED_F32_EXEMPT):
                bad.append("%s: %s" % (func, ln.strip()))
    if bad:
        raise RuntimeError("fused f32 multiply-add in device code (breaks bit-exact parity):\n" + "\n".join(bad[:10]))
    # the pair-test kernel runs five waves per SIMD only up to 96 vector registers; one more costs a wave (3 % of its time)
    with open(asm) as f:
        text = f.read()
    for kernel, limit in (("pc_c3_count_groups", 96), ("pc_c5_count_merged", 80)):
        for m in re.finditer(r"\.set (_Z\d+%s\w*)\.num_vgpr, (\d+)" % kernel, text):
            if int(m.group(2)) > limit:
                print("warning: %s uses %s vector registers (> %d: one wave per SIMD fewer)" % (m.group(1), m.group(2), limit),
                      file=sys.stderr)
            spill = re.search(r"\.set %s\.private_seg_size, (\d+)" % re.escape(m.group(1)), text)
            if spill and int(spill.group(1)) > 0:
                print("warning: %s spills %s bytes per lane" % (m.group(1), spill.group(1)), file=sys.stderr)
    _run([HIPCC] + HIP_FLAGS + ["-shared", "-o", out, main_src, "-lpthread"])
    return out


def build_host(force=False):
    outs = []
    if not os.path.isdir(HOST):
        return outs
    hdrs = _sources(HOST, (".h",)) + [os.path.join(ROOT, "include", "paircount.h")]
    gen = os.path.join(HOST, "xyzhost.c")
    if os.path.exists(gen):
        out = os.path.join(HERE, "libxyzhost.so")
        if force or _newer(out, [gen] + hdrs):
            _run([CC] + C_FLAGS + ["-shared", "-o", out, gen])
        outs.append(out)
    main_c = os.path.join(HOST, "main.c")
    if os.path.exists(main_c):
        out = os.path.join(HERE, "main")
        if force or _newer(out, [main_c] + hdrs + [os.path.join(HERE, "libpaircount.so")]):
            _run([CC] + C_FLAGS + ["-o", out, main_c, "-L" + HERE, "-lpaircount", "-Wl,-rpath,$ORIGIN"])
        outs.append(out)
    return outs


def build_oracle(force=False):
    """The CPU checker (test infrastructure).  Building it is not using it."""
    odir = os.path.join(ROOT, "oracle")
    _run(["make", "-C", odir] + (["-B"] if force else []))
    return os.path.join(odir, "liboracle.so")


def build_sanitizer_flavours(force=False):
    """The host code under AddressSanitizer+UndefinedBehaviorSanitizer and under ThreadSanitizer (CPU only, test
    infrastructure: oracle/Makefile target `san`; the reference's counterpart is `make debug`, Makefile:12-17)."""
    odir = os.path.join(ROOT, "oracle")
    _run(["make", "-C", odir, "san"] + (["-B"] if force else []))
    return [os.path.join(odir, "_san", n) for n in ("main_asan", "main_tsan", "xyzhost_asan", "xyzhost_tsan")]


def build_all(force=False):
    lib = build_hip(force)
    host = build_host(force)
    orc = build_oracle(force)
    return [lib] + host + [orc]


if __name__ == "__main__":
    for p in build_all("--force" in sys.argv) + (build_sanitizer_flavours("--force" in sys.argv) if "--sanitize" in sys.argv else []):
        print("built", os.path.relpath(p, ROOT))
