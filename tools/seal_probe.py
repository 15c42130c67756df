#!/usr/bin/env python
"""Looks for an installed Microsoft SEAL >= 3.6 on this host ("if
find_package(SEAL 3.6) succeeds on the GPU host, link it, feed identical keys / inputs and diff the
ciphertexts bit for bit; absence is reported, not hidden").

  probe()            -> {"present": bool, "how": "...", "paths": [...]}
  build_and_run(dir) -> when present: builds tools/seal_parity (cmake) and runs it on the exported
                        vectors; returns its summary; {"present": False, ...} otherwise.
Never required: the product does not link SEAL.  Used by bench.py's cpu_baseline leg and by
tests/test_seal_parity.py."""
import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def probe():
    found = []
    try:
        out = subprocess.run(["ldconfig", "-p"], capture_output=True, text=True, timeout=20).stdout
        found += [ln.split("=>")[-1].strip() for ln in out.splitlines() if "libseal" in ln.lower()]
    except Exception:
        pass
    for pat in ("/usr/lib*/cmake/SEAL*", "/usr/local/lib*/cmake/SEAL*", "/opt/*/lib*/cmake/SEAL*",
                "/usr/include/SEAL*", "/usr/local/include/SEAL*", "/usr/lib/x86_64-linux-gnu/cmake/SEAL*",
                os.path.expanduser("~/.local/lib*/cmake/SEAL*")):
        found += glob.glob(pat)
    for var in ("SEAL_DIR", "SEAL_ROOT"):
        if os.environ.get(var) and os.path.exists(os.environ[var]):
            found.append(os.environ[var])
    return {"present": bool(found), "how": "ldconfig -p | grep seal; cmake package dirs; $SEAL_DIR", "paths": found}


def build_and_run(vector_dir, time_triple=False):
    p = probe()
    if not p["present"] or not shutil.which("cmake"):
        return {"present": False, "reason": "Microsoft SEAL >= 3.6 not installed on this host" if not p["present"] else "cmake missing"}
    bdir = os.path.join(ROOT, "build", "seal_parity")
    os.makedirs(bdir, exist_ok=True)
    env = dict(os.environ)
    r = subprocess.run(["cmake", "-S", HERE, "-B", bdir, "-DCMAKE_BUILD_TYPE=Release"], capture_output=True, text=True, env=env)
    if r.returncode:
        return {"present": True, "built": False, "log": (r.stdout + r.stderr)[-2000:]}
    r = subprocess.run(["cmake", "--build", bdir, "-j", "8"], capture_output=True, text=True)
    if r.returncode:
        return {"present": True, "built": False, "log": (r.stdout + r.stderr)[-2000:]}
    cmd = [os.path.join(bdir, "seal_parity"), vector_dir] + (["--time-triple"] if time_triple else [])
    r = subprocess.run(cmd, capture_output=True, text=True)
    lines = r.stdout.splitlines()
    timing = next((ln for ln in lines if ln.startswith("TIMING")), None)
    return {"present": True, "built": True, "ok": r.returncode == 0, "summary": lines[-1] if lines else "",
            "failed": [ln for ln in lines if ln.startswith("FAIL")], "timing": timing, "paths": p["paths"]}


if __name__ == "__main__":
    import json
    print(json.dumps(probe()))
