"""Where does the wall time of one resident Harris execute() go: host time until execute() returns (plan lookup, slot
refill, graph launch, output copies), then the wait for the GPU.  python scripts/harris_latency_probe.py (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
from eva.ckks import CKKSCompiler
from eva.seal import generate_keys
from eva_amd.workloads import harris as _harris, image as _image
import bench
compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(_harris())
bench.pad_chain(params, 9, 32768)
pub, sec = generate_keys(params, 1)
enc = pub.encrypt(_image(4096), sig)
for _ in range(5):
    out = pub.execute(compiled, enc)
pub.synchronize()
host, total, parts = [], [], []
for _ in range(40):
    t0 = time.perf_counter()
    out = pub.execute(compiled, enc)
    t1 = time.perf_counter()
    pub.synchronize()
    t2 = time.perf_counter()
    host.append(t1 - t0); total.append(t2 - t0); parts.append(list(pub.last_timing))
med = lambda xs: sorted(xs)[len(xs) // 2]
print(f"execute() returns after {med(host)*1e6:7.1f} us; execute + synchronize {med(total)*1e6:7.1f} us; "
      f"C++ run_plan parts (refill, graph launch, outputs) {[round(med([p[j] for p in parts])*1e3, 1) for j in range(3)]} us")
# back-to-back replays without a wait in between: the GPU-side time per replay
t0 = time.perf_counter()
for _ in range(100):
    out = pub.execute(compiled, enc)
pub.synchronize()
print(f"100 replays back to back: {(time.perf_counter() - t0) / 100 * 1e6:7.1f} us per replay")
