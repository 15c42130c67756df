"""Run one DAG config repeatedly (graph replay) for rocprofv3 --kernel-trace: dag_profile.py sobel|harris [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from eva.ckks import CKKSCompiler
from eva.seal import generate_keys
from test_compiler import _sobel
from test_gpu_e2e import _harris, _image
which = sys.argv[1] if len(sys.argv) > 1 else "sobel"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if which == "sobel":
    prog = _sobel(64, 64, 4096); prog.set_input_scales(25); prog.set_output_ranges(10); N = 8192
else:
    prog = _harris(); N = 32768
compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
params.poly_modulus_degree = N
pub, sec = generate_keys(params, 1)
enc = pub.encrypt(_image(4096), sig)
for _ in range(reps + 2):
    pub.execute(compiled, enc)
