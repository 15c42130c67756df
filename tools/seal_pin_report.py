#!/usr/bin/env python
"""tools/pin_with_seal.sh, last step: the logs of tools/seal_parity (one per exported vector set) -> one JSON,
profiles/seal_pin.json by default:

  {"sections": {"<N>:<bits>": {"passed": n, "failed": ["FAIL ..."], "summary": "SUMMARY ..."}}, "all_identical": bool,
   "seal_triples_per_s": float | null, "N": .., "limbs": .., "cores": 1, "seal_version": "..", "host": "..", "sample": ".."}

bench.py reads it (seal_pin_baseline): when it holds an op-triple rate at the bench's (N, L) the reported cpu_baseline is
the reference itself (kind "reference") and the oracle port's figures move beside it.  --dry-run writes a report that says
no SEAL took part (sections null) — it can never be mistaken for a pin.  Test infrastructure / tooling: not product code."""
import argparse
import json
import os
import re
import socket
import sys


def manifest(d):
    path = os.path.join(d, "manifest.txt")
    if not os.path.exists(path):
        return {}
    return dict(ln.split(" ", 1) for ln in open(path).read().splitlines() if ln and not ln.startswith("#") and " " in ln)


def parse_log(text):
    lines = text.splitlines()
    passed = sum(1 for ln in lines if ln.startswith("PASS"))
    failed = [ln for ln in lines if ln.startswith("FAIL")]
    summary = next((ln for ln in reversed(lines) if ln.startswith("SUMMARY")), "")
    timing = next((ln for ln in lines if ln.startswith("TIMING")), None)
    rate = None
    if timing:
        m = re.match(r"TIMING op-triples/s ([0-9.eE+-]+) threads (\d+) N (\d+) limbs (\d+)", timing)
        if m:
            rate = {"rate": float(m.group(1)), "threads": int(m.group(2)), "N": int(m.group(3)), "limbs": int(m.group(4))}
    complete = bool(summary) and f"{passed} passed" in summary
    return {"passed": passed, "failed": failed, "summary": summary, "complete": complete}, rate


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--out", required=True)
    ap.add_argument("--dry-run", action="store_true")
    ap.add_argument("--cmake-log")
    a = ap.parse_args(argv)
    rep = {"host": socket.gethostname(), "tool": "tools/pin_with_seal.sh", "sections": None, "all_identical": None,
           "seal_triples_per_s": None, "vector_sets": []}
    for d in a.dirs:
        man = manifest(d)
        rep["vector_sets"].append({"dir": os.path.basename(d), "N": int(man.get("N", 0) or 0), "bits": man.get("bits", "")})
    if a.dry_run:
        rep["dry_run"] = True
        rep["note"] = "no Microsoft SEAL took part: vectors exported and the checker type-checked against declarations only"
    else:
        rep["sections"] = {}
        ok = True
        for d in a.dirs:
            man = manifest(d)
            key = f"{man.get('N', '?')}:{man.get('bits', '?').replace(' ', ',')}"
            log = os.path.join(d, "seal_parity.log")
            if not os.path.exists(log):
                rep["sections"][key] = {"passed": 0, "failed": ["seal_parity did not run"], "summary": "", "complete": False}
                ok = False
                continue
            sec, rate = parse_log(open(log).read())
            rep["sections"][key] = sec
            ok = ok and sec["complete"] and not sec["failed"] and sec["passed"] > 0
            if rate:
                rep.update({"seal_triples_per_s": rate["rate"], "cores": rate["threads"], "N": rate["N"], "limbs": rate["limbs"],
                            "sample": "20 x (Evaluator::multiply, relinearize, rescale_to_next) on one thread, tools/seal_parity.cpp --time-triple"})
        rep["all_identical"] = ok
        if a.cmake_log and os.path.exists(a.cmake_log):
            m = re.search(r"SEAL[^\n]*?(\d+\.\d+\.\d+)", open(a.cmake_log).read())
            rep["seal_version"] = m.group(1) if m else "?"
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(rep, f, indent=1)
        f.write("\n")
    print(json.dumps({k: rep[k] for k in ("all_identical", "seal_triples_per_s", "sections") if k in rep})[:600])
    return 0 if (a.dry_run or rep["all_identical"]) else 1


if __name__ == "__main__":
    sys.exit(main())
