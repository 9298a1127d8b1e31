"""The drop-in boundary is a C ABI: the header must be valid C99 and a C program must link and run against the library."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "abi_check.c")
LIBDIR = os.path.join(ROOT, "crabml_amd")


def _build(tmp_path):
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no C compiler")
    exe = str(tmp_path / "abi_check")
    subprocess.check_call([gcc, "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), SRC,
                           "-L", LIBDIR, "-lcrabml_amd", "-lm", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_header_is_c99_and_a_c_program_links_and_runs(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, f"abi_check exit {out.returncode}: {out.stdout} {out.stderr}"
    assert out.stdout.startswith("abi ok")


@pytest.mark.gpu
def test_c_program_evaluates_on_the_device(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, f"abi_check exit {out.returncode}: {out.stdout} {out.stderr}"
    assert "device: lnL" in out.stdout


def test_the_constant_in_the_c_program_is_the_oracles_value():
    """abi_check.c compares the device's lnL of its 3-leaf case with a constant (1e-10 relative): recompute it with the oracle."""
    import re
    import sys
    import ctypes as C
    import numpy as np
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    from crabml_amd import _lib as L
    lib = L.load()
    pi = np.full(61, 1.0 / 61.0)
    Q = np.empty((61, 61))
    L.check(lib.crab_build_codon_q_matrix(C.c_double(2.0), C.c_double(0.5), pi.ctypes.data_as(L.dp), 0, C.c_double(0.0), Q.ctypes.data_as(L.dp)))
    parents = [3, 3, 4, 4, -1]
    ts = [(i, (p if p >= 0 else None)) for i, p in enumerate(parents)]
    codes = np.array([[0, 5], [0, 5], [1, 64]], dtype=np.int32)
    ref = orc.compute_site_class_log_likelihood([Q], [1.0], pi, ts, [0.1, 0.2, 0.3, 0.1, 0.0], None, codes, [0, 1, 2])
    const = float(re.search(r"lnL_oracle = (-?[0-9.]+);", open(SRC).read()).group(1))
    assert abs(const - ref) <= 1e-13 * abs(ref)
