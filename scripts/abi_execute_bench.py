"""The C-ABI one-call submit (evah_execute) on the BASELINE DAGs, driven from Python through ctypes
the way a C++ maintainer would drive it: eager call and hipGraph replay (capture around the call).
usage: abi_execute_bench.py [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from eva.ckks import CKKSCompiler
from eva.seal import generate_keys
from eva_amd import backend as be
from eva_amd.workloads import sobel as _sobel
from eva_amd.workloads import harris as _harris, image as _image
from test_gpu_execute_abi import _lower

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def run(name, prog, N):
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    params.poly_modulus_degree = N
    pub, sec = generate_keys(params, 1)
    g = be.Context(N, list(pub.primes))
    g.upload_relin_key(pub.relin_key())
    for elt, key in pub.galois_keys().items():
        g.upload_galois_key(elt, key)
    enc = pub.encrypt(_image(4096), sig)
    ops, values, outs = _lower(compiled, enc, pub, g)

    def once():
        res = g.execute(ops, dict(values))
        for t, h in res.items():
            if t not in values:
                h.free()
    once(); once()
    g.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    g.sync()
    eager = (time.perf_counter() - t0) / reps
    g.capture_begin()
    res = g.execute(ops, dict(values))
    graph = g.capture_end()
    g.graph_launch(graph); g.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.graph_launch(graph)
    g.sync()
    replay = (time.perf_counter() - t0) / reps
    print(f"{name}: N={N}, {len(ops)} encrypted ops -> evah_execute eager {eager*1e3:.2f} ms, captured + replayed {replay*1e3:.2f} ms "
          f"(inputs resident, no download)", flush=True)
    g.graph_free(graph)


sob = _sobel(64, 64, 4096); sob.set_input_scales(25); sob.set_output_ranges(10)
run("C2 sobel", sob, 8192)
run("C3 harris", _harris(), 32768)
