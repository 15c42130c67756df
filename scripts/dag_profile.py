"""Run one DAG config repeatedly (graph replay) for rocprofv3 --kernel-trace: dag_profile.py sobel|harris|harris8|deep [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from eva.ckks import CKKSCompiler
from eva.seal import generate_keys
from eva_amd.workloads import sobel as _sobel
from eva_amd.workloads import harris as _harris, image as _image
which = sys.argv[1] if len(sys.argv) > 1 else "sobel"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if which == "sobel":
    prog = _sobel(64, 64, 4096); prog.set_input_scales(25); prog.set_output_ranges(10); N = 8192
elif which in ("harris", "harris8"):
    prog = _harris(); N = 32768
else:  # BASELINE config 5: 3x3 convolution + depth-8 squaring chain, N = 2^16, 13 primes
    from eva import EvaProgram, Input, Output
    prog = EvaProgram('conv+depth8', vec_size=4096)
    with prog:
        image = Input('image')
        acc = None
        for i in range(3):
            for j in range(3):
                t = (image << (i * 64 + j)) * (1.0 / 9.0)
                acc = t if acc is None else acc + t
        for _ in range(8):
            acc = acc * acc
        Output('y', acc)
    prog.set_input_scales(30); prog.set_output_ranges(20); N = 65536
compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
params.poly_modulus_degree = N
if which == "harris8":  # the bench's DAG leg: chain padded to L = 8 data limbs
    pb = list(params.prime_bits)
    params.prime_bits = pb[:1] + [60] * (9 - len(pb)) + pb[1:]
if which == "deep" and len(params.prime_bits) < 13:
    pb = list(params.prime_bits)
    params.prime_bits = pb[:1] + [60] * (13 - len(pb)) + pb[1:]
pub, sec = generate_keys(params, 1)
enc = pub.encrypt(_image(4096), sig)
for _ in range(reps + 2):
    pub.execute(compiled, enc)
pub.synchronize()
