"""config 5 execute(): hipGraph replay vs eager walk, with / without the side stream of the fused Mul -> Rescale -> Relinearize (r6)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from eva.seal import generate_keys
from eva_amd import workloads
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
compiled, params, sig, inputs = workloads.compile_config(which)
for graphs in (True, False):
    pub, sec = generate_keys(params, 1)
    pub.use_graphs = graphs
    enc = pub.encrypt(inputs, sig)
    for _ in range(4):
        out = pub.execute(compiled, enc)
    pub.synchronize()
    ts = []
    for _ in range(15):
        t0 = time.perf_counter(); out = pub.execute(compiled, enc); pub.synchronize(); ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    for _ in range(30):
        out = pub.execute(compiled, enc)
    pub.synchronize()
    b2b = (time.perf_counter() - t0) / 30
    print(f"{which} graphs={graphs} side={os.environ.get('EVAH_SIDE_STREAM','1')} fuse2={os.environ.get('EVAH_FUSE_MUL2','1')}: resident median {sorted(ts)[7]*1e3:.3f} ms  back-to-back {b2b*1e3:.3f} ms")
    del out, enc, pub, sec
