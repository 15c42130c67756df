"""EVAH_EW_DEBUG=1 python scripts/ew_debug_probe.py [batch]: the elementwise programs evah_execute builds for Sobel (config 2 shape), one
execute() and one execute_batch group — what each launch of k_ew_program holds"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eva.seal import generate_keys  # noqa: E402
from eva_amd import workloads  # noqa: E402

compiled, params, sig, inputs = workloads.compile_config("c2")
pub, sec = generate_keys(params, 1)
enc = pub.encrypt(inputs, sig)
print("---- eager walk, one instance", file=sys.stderr)
out = pub.execute(compiled, enc)
pub.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pub.batch_chunk = n
print(f"---- execute_batch, one group of {n}", file=sys.stderr)
outs = pub.execute_batch(compiled, [enc] * n)
pub.synchronize()
