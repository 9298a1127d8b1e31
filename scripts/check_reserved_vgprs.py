"""Checks the generated ISA of the pruning kernels for the two things the compiler does not guarantee around inline asm:
(1) outside the inline-asm blocks (which name the fixed tile registers on purpose) the compiler must not touch any VGPR at or
    above the limit of the kernel's register plan;
(2) a vector-memory instruction inside an asm block must not read a scalar register that a VALU instruction (v_readlane,
    v_readfirstlane, a VALU compare) wrote fewer than five wait states earlier: the hazard recognizer inserts those wait states
    for the compiler's own instructions only.
usage: check_reserved_vgprs.py [extra hipcc flags...]   (exit status 1 on violation)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "crabml_amd", "csrc", "crab_kernels.hip")
LIMITS = {"prune_kernel_nt1": 64, "prune_kernel_nt2": 96}


def violations(extra=()):
    limits = dict(LIMITS)
    if "-DCRAB_PROF" in extra:
        limits["prune_kernel_nt1"] = 96   # Plan<1>::CC of that build
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                               *extra, SRC, "-o", out], stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    bad = []
    cur, in_asm = None, False
    for ln, line in enumerate(text, 1):
        m = re.match(r"^(_ZN4crab\d+(prune_kernel_nt\d)\w*):", line)
        if m:
            cur = m.group(2)
        if "s_endpgm" in line:
            cur = None
        if ";;#ASMSTART" in line:
            in_asm = True
            continue
        if ";;#ASMEND" in line:
            in_asm = False
            continue
        if cur is None or in_asm or line.lstrip().startswith((";", ".")):
            continue
        code = line.split(";")[0]
        regs = [int(x) for x in re.findall(r"\bv(\d+)\b", code)]
        for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", code):
            regs += [int(a), int(b)]
        if regs and max(regs) >= limits[cur]:
            bad.append((cur, ln, line.strip()))
    return bad + sgpr_hazards(text)


def sgpr_hazards(text):
    """(2) above, over the pruning kernels' code."""
    ins = []          # (kernel, line number, text, inside asm)
    cur, in_asm = None, False
    for ln, line in enumerate(text, 1):
        m = re.match(r"^(_ZN4crab\d+(prune_kernel_nt\d)\w*):", line)
        if m:
            cur = m.group(2)
        if "s_endpgm" in line:
            cur = None
        if ";;#ASMSTART" in line:
            in_asm = True
            continue
        if ";;#ASMEND" in line:
            in_asm = False
            continue
        code = line.split(";")[0].strip()
        if cur is None or not code or code.startswith(".") or code.endswith(":"):
            continue
        ins.append((cur, ln, code, in_asm))

    def sregs(tok):
        m = re.fullmatch(r"s(\d+)", tok)
        if m:
            return {int(m.group(1))}
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
        return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()

    bad = []
    for k, (kern, ln, code, in_asm) in enumerate(ins):
        if not in_asm or not re.match(r"(global_|buffer_|flat_|scratch_)", code):
            continue
        need = set()
        for tok in re.split(r"[ ,]+", code)[1:]:
            need |= sregs(tok)
        ws, j = 0, k - 1
        while need and j >= 0 and ws < 5 and ins[j][0] == kern:
            prev = ins[j][2]
            m = re.match(r"s_nop (\d+)", prev)
            if m:
                ws += int(m.group(1)) + 1
                j -= 1
                continue
            toks = re.split(r"[ ,]+", prev)
            dst = sregs(toks[1]) if len(toks) > 1 else set()
            hit = dst & need
            if hit:
                if prev.startswith("v_"):
                    bad.append((kern, ln, code + "   <- " + prev))
                need -= hit   # written here (by a scalar instruction: no hazard); earlier writers are irrelevant
            ws += 1
            j -= 1
    return bad


if __name__ == "__main__":
    bad = violations(sys.argv[1:])
    for k, ln, line in bad[:20]:
        print(f"{k}: line {ln}: {line}")
    print(f"{len(bad)} compiler-generated instruction(s) touch reserved registers")
    sys.exit(1 if bad else 0)
