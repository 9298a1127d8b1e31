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
