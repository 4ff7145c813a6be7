"""GPU: the C ABI driven from a plain C++ program (HIP runtime + libfdm_hip.so, no Python objects, no torch):
tests/abi_c/abi_smoke.cpp is compiled with hipcc against include/fdm_hip.h and run as a child process."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_c_program_links_and_runs_against_the_library(tmp_path):
    from fdm_amd import _lib
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "abi_smoke")
    cmd = [hipcc, "-O2", "--offload-arch=gfx950", os.path.join(ROOT, "tests", "abi_c", "abi_smoke.cpp"),
           "-I", os.path.join(ROOT, "include"), "-L", libdir, "-lfdm_hip", "-Wl,-rpath," + libdir, "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi_smoke ok" in r.stdout
