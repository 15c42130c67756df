"""examples/c_abi_triple.c: include/eva_hip.h is a C header and libeva_hip.so links from plain C
(no C++ / torch types at the boundary); on the GPU the example's per-op path and its evah_execute
path must agree."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "c_abi_triple")


def _build():
    cmd = ["gcc", "-O2", "-Wall", "-Werror", "-std=c11", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "c_abi_triple.c"), "-L", os.path.join(ROOT, "eva_amd", "lib"), "-leva_hip",
           "-Wl,-rpath," + os.path.join(ROOT, "eva_amd", "lib"), "-lm", "-o", EXE]
    subprocess.check_call(cmd)


def test_header_compiles_and_links_as_c():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_c_example_runs_and_both_paths_agree():
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "evah_execute match" in out.stdout
