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


@pytest.mark.gpu
def test_python_examples_run(tmp_path):
    """examples/image_filters.py (Sobel, Harris) and examples/client_server_files.py end to end"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    for args in (["examples/image_filters.py", "sobel", "--size", "32", "--pgm", str(tmp_path / "s.pgm")],
                 ["examples/image_filters.py", "harris", "--size", "32"],
                 ["examples/client_server_files.py", str(tmp_path)]):
        out = subprocess.run([sys.executable] + args, cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        mse = float(out.stdout.split("MSE vs")[1].split(":")[1].split()[0])
        assert mse < 0.01, out.stdout
    assert (tmp_path / "s.pgm").stat().st_size > 32 * 32
