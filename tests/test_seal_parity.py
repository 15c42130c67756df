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
    # section 7 of the checker: every SEAL object it loads was written (by this repo's writer of SEAL's format),
    # begins with SEAL's header and says how long it is
    import struct
    for name in re.findall(r'load_bytes\("([a-z0-9_]+)"\)', src):
        blob = open(os.path.join(out, name + ".bin"), "rb").read()
        magic, hsize, major, minor, compr, _, size = struct.unpack_from("<HBBBBHQ", blob)
        assert (magic, hsize, major, minor, compr, size) == (0xA15E, 16, 3, 6, 0, len(blob)), name
    assert "pk.u64" in files


def test_checker_compiles_against_the_declared_seal_api():
    """tools/seal_parity.cpp has never met a real SEAL where this repository is developed: at least it must
    parse and type-check against the SEAL 3.6 declarations it uses (tools/seal_api_stub: declarations only,
    compile-check only — it pins nothing and cannot be linked)."""
    import shutil
    if not shutil.which("g++"):
        pytest.skip("no g++")
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "tools", "seal_api_stub"),
                        os.path.join(ROOT, "tools", "seal_parity.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    stub = open(os.path.join(ROOT, "tools", "seal_api_stub", "seal", "seal.h")).read()
    assert "COMPILE-CHECK ONLY" in stub


def test_fresh_vector_sets_cover_the_cases_the_verdicts_named(tmp_path):
    """N = 2^13 with one of EVA's 20-bit output primes, a size-3 rescale, a negative rotation, an encode at
    2^60, decrypt and decode: all in one exported set (what a SEAL-equipped host replays)."""
    out = str(tmp_path / "vec8192")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "golden", "export_seal_vectors.py"), out, "8192", "60,20,60,60"])
    files = set(os.listdir(out))
    for name in ("out_rescale3.u64", "out_rotate_-3.u64", "out_encode_4.u64", "out_decrypt3.u64", "out_decode_pt.f64", "sk_ntt.u64"):
        assert name in files, name
    man = dict(ln.split(" ", 1) for ln in open(os.path.join(out, "manifest.txt")).read().splitlines() if not ln.startswith("#"))
    assert man["N"] == "8192" and man["bits"] == "60 20 60 60" and man["enc_scale_bits"].split()[-1] == "60"
    primes = np.fromfile(os.path.join(out, "primes.u64"), dtype="<u8")
    assert int(primes[1]) == 0xFC001  # the 20-bit prime of (8192, [60, 20, 60, 60])


def test_real_seal_reproduces_the_golden_vectors(tmp_path):
    res = seal_probe.build_and_run(_export(tmp_path))
    if not res["present"]:
        pytest.skip("SEAL absent: " + res["reason"] + " — parity stays pinned to the oracle only")
    assert res.get("built"), res.get("log")
    assert res["ok"], res["failed"]
