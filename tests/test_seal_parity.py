"""The recipe by which `parity` is pinned to Microsoft SEAL's own bits:
tests/golden/export_seal_vectors.py writes the golden vectors as raw files, tools/seal_parity.cpp
replays them through a real SEAL >= 3.6 and compares every output word.  Here: the exporter works
and is complete; the checker's source names every exported vector; and, when SEAL is installed on
the host, the checker is built and must pass (otherwise the absence is reported as a skip)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import seal_probe  # noqa: E402


def _export(tmp_path):
    out = str(tmp_path / "vec")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "golden", "export_seal_vectors.py"), out])
    return out


def test_exporter_writes_every_vector_the_checker_reads(tmp_path):
    out = _export(tmp_path)
    files = set(os.listdir(out))
    src = open(os.path.join(ROOT, "tools", "seal_parity.cpp")).read()
    for name in re.findall(r'load_u64\("([a-z0-9_]+)"\)', src):
        assert name + ".u64" in files, name
    vec = np.load(os.path.join(ROOT, "tests", "golden", "ops_n1024.npz"))
    for s in vec["rot_steps"]:
        assert f"galois_key_{int(s)}.u64" in files and f"out_rotate_{int(s)}.u64" in files
    for c in range(len(vec["enc_scale_bits"])):
        assert f"enc_values_{c}.f64" in files and f"out_encode_{c}.u64" in files
    back = np.fromfile(os.path.join(out, "out_triple.u64"), dtype="<u8").reshape(vec["out_triple"].shape)
    assert np.array_equal(back, vec["out_triple"])
    man = dict(ln.split(" ", 1) for ln in open(os.path.join(out, "manifest.txt")).read().splitlines() if not ln.startswith("#"))
    assert man["N"] == "1024" and man["bits"] == "60 40 60"


def test_real_seal_reproduces_the_golden_vectors(tmp_path):
    res = seal_probe.build_and_run(_export(tmp_path))
    if not res["present"]:
        pytest.skip("SEAL absent: " + res["reason"] + " — parity stays pinned to the oracle only")
    assert res.get("built"), res.get("log")
    assert res["ok"], res["failed"]
