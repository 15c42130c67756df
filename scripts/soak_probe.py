"""Soak: thousands of Harris replays, op-triple execute() calls and config-4 batches in one process; device memory in use
(hipMemGetInfo) must stop growing after the first iterations.  python scripts/soak_probe.py  (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import numpy as np
from eva.ckks import CKKSCompiler
from eva.seal import generate_keys
from eva_amd.workloads import harris as _harris, image as _image
import bench


def used_mb():
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2 ** 20


compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(_harris())
bench.pad_chain(params, 9, 32768)
pub, sec = generate_keys(params, 1)
enc = pub.encrypt(_image(4096), sig)
ref = None
marks = []
for it in range(4000):
    out = pub.execute(compiled, enc)
    if it in (10, 100, 1000, 3999):
        pub.synchronize()
        w = out.get('image')[4] if 'image' in out.names() else out.get(out.names()[0])[4]
        ref = w if ref is None else ref
        assert np.array_equal(w, ref)
        marks.append((it, round(used_mb(), 1)))
print("harris replays:", marks, flush=True)
tc, tp, ts = bench.triple_program(32, 65536, 10)
pub2, sec2 = generate_keys(tp, 17)
rng = np.random.default_rng(0)
v = pub2.encrypt({**{f'x{i}': list(rng.uniform(-1, 1, 1024)) for i in range(32)}, **{f'y{i}': list(rng.uniform(-1, 1, 1024)) for i in range(32)}}, ts)
marks = []
for it in range(600):
    o = pub2.execute(tc, v)
    if it in (5, 50, 599):
        pub2.synchronize()
        marks.append((it, round(used_mb(), 1)))
print("op-triple execute() calls:", marks, flush=True)
state = bench._dag_batch_setup(256, 0, 1, 0, 1)
pubb, _, cb, _, _, inputs, _ = state
marks = []
for it in range(60):
    outs = pubb.execute_batch(cb, inputs)
    if it in (2, 20, 59):
        marks.append((it, round(used_mb(), 1)))
print("execute_batch calls:", marks, flush=True)
