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


def test_pin_kit_one_command(tmp_path, monkeypatch):
    """tools/pin_with_seal.sh is the one command a SEAL-equipped host runs.  Here, without SEAL: its --dry-run (vector
    export for a golden and a fresh set + the checker type-checked against the declarations) must work end to end and
    say that it pinned nothing; its report step must turn seal_parity logs into the JSON bench.py ingests, and bench.py
    must then report the reference as the baseline — and must NOT do so for a dry run or for another (N, L)."""
    import json
    import shutil
    if not shutil.which("bash") or not shutil.which("g++"):
        pytest.skip("no bash / g++")
    script = os.path.join(ROOT, "tools", "pin_with_seal.sh")
    assert subprocess.run(["bash", "-n", script]).returncode == 0
    out = tmp_path / "pin.json"
    env = dict(os.environ, PIN_CONFIGS="golden 2048:40,20,40,41", PIN_JSON=str(out), PYTHON=sys.executable)
    r = subprocess.run(["bash", script, "--dry-run", str(tmp_path / "work")], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = json.loads(out.read_text())
    assert rep["dry_run"] is True and rep["sections"] is None and rep["all_identical"] is None and rep["seal_triples_per_s"] is None
    assert [v["N"] for v in rep["vector_sets"]] == [1024, 2048]
    # the report step on logs as tools/seal_parity.cpp prints them (report(): "PASS  <what>" / "FAIL  <what>", TIMING, SUMMARY)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import seal_pin_report
    dirs = [str(tmp_path / "work" / "vec_golden"), str(tmp_path / "work" / "vec_2048")]
    open(os.path.join(dirs[0], "seal_parity.log"), "w").write("PASS  CoeffModulus::Create\nPASS  psi\nSUMMARY 2 passed, 0 failed\n")
    open(os.path.join(dirs[1], "seal_parity.log"), "w").write(
        "PASS  multiply\nPASS  relinearize\nPASS  rescale_to_next\nTIMING op-triples/s 12.5 threads 1 N 65536 limbs 10\nSUMMARY 3 passed, 0 failed\n")
    assert seal_pin_report.main(["--out", str(out)] + dirs) == 0
    rep = json.loads(out.read_text())
    assert rep["all_identical"] is True and rep["seal_triples_per_s"] == 12.5 and (rep["N"], rep["limbs"]) == (65536, 10)
    assert rep["sections"]["1024:60,40,60"]["passed"] == 2 and rep["sections"]["2048:40,20,40,41"]["failed"] == []
    # one FAIL line (or a log without its SUMMARY) is not "identical"
    open(os.path.join(dirs[0], "seal_parity.log"), "w").write("PASS  psi\nFAIL  ntt_negacyclic_harvey\nSUMMARY 1 passed, 1 failed\n")
    bad = tmp_path / "bad.json"
    assert seal_pin_report.main(["--out", str(bad)] + dirs) == 1 and json.loads(bad.read_text())["all_identical"] is False
    # bench.py: the reference becomes the reported baseline only for a real report at the bench's own (N, L)
    sys.path.insert(0, ROOT)
    import bench
    port = {"value": 8.9, "unit": "op-triples/s", "cores": 1, "kind": "port", "sample": "oracle", "all_cores": {"value": 70.0}, "threads_64": None,
            "seal": "SEAL absent"}
    monkeypatch.setenv("EVA_SEAL_PIN_JSON", str(out))
    got = bench.seal_pin_baseline(dict(port), 65536, 10)
    assert got["kind"] == "reference" and got["value"] == 12.5 and got["port"]["value"] == 8.9 and got["seal_pin"]["all_sections_identical"] is True
    assert bench.seal_pin_baseline(dict(port), 32768, 8)["kind"] == "port"
    out.write_text(json.dumps({"dry_run": True, "sections": None, "seal_triples_per_s": None}))
    again = bench.seal_pin_baseline(dict(port), 65536, 10)
    assert again["kind"] == "port" and again["seal_pin"]["used"] is False
