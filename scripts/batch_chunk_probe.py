"""config 4 (256 Sobel DAGs, N = 2^14, l = 5) through execute_batch with different group sizes / queue depths:
python scripts/batch_chunk_probe.py   (run on the GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import bench
state = bench._dag_batch_setup(256, 0, 1, 0, 1)
pub, sec, compiled, params, nbytes, inputs, mine = state
import gc
for chunk, depth in ((16, 4), (8, 4), (16, 6), (16, 8), (8, 8), (12, 4), (24, 4), (16, 4), (32, 4)):
    pub.batch_chunk, pub.batch_depth = chunk, depth
    for _ in range(3):
        pub.execute_batch(compiled, inputs)
    gc.collect(); gc.disable()
    ts = []
    for _ in range(7):
        outs = None
        t0 = time.perf_counter()
        outs = pub.execute_batch(compiled, inputs)
        ts.append(time.perf_counter() - t0)
    gc.enable()
    ts.sort()
    print(f"chunk {chunk:3d} depth {depth}: median {ts[3]*1e3:6.2f} ms = {256/ts[3]:8.1f} DAGs/s  (best {ts[0]*1e3:.2f} ms)", flush=True)
