"""Checks the register plan of the pruning kernels: outside the inline-asm blocks (which name the fixed tile registers on
purpose) the compiler must not touch any VGPR at or above the limit given by amdgpu_num_vgpr.
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
    return bad


if __name__ == "__main__":
    bad = violations(sys.argv[1:])
    for k, ln, line in bad[:20]:
        print(f"{k}: line {ln}: {line}")
    print(f"{len(bad)} compiler-generated instruction(s) touch reserved registers")
    sys.exit(1 if bad else 0)
