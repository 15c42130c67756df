"""probe: Harris / Sobel execute() (graph replay) under the current environment; prints min / median ms"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from eva.ckks import CKKSCompiler
from eva.seal import generate_keys
from eva_amd.workloads import sobel as _sobel
from eva_amd.workloads import harris as _harris, image as _image
for name, prog, N in (("sobel", None, 8192), ("harris", _harris(), 32768)):
    if prog is None:
        prog = _sobel(64, 64, 4096); prog.set_input_scales(25); prog.set_output_ranges(10)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
    params.poly_modulus_degree = N
    pub, sec = generate_keys(params, 1)
    enc = pub.encrypt(_image(4096), sig)
    for _ in range(3):
        pub.execute(compiled, enc)
    pub.synchronize()
    ts = []
    for _ in range(15):
        t0 = time.perf_counter(); pub.execute(compiled, enc); pub.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"{name}: min {ts[0]*1e3:.3f} median {ts[7]*1e3:.3f} ms  timing {['%.3f' % x for x in pub.last_timing]}", flush=True)
