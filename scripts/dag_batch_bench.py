"""BASELINE config 4: a batch of independent Sobel DAGs (N forced to 2^14) sharded over the GPUs of
one node, one process per GPU, no data-path collective.  Launch like bench.py:
  python scripts/dag_batch_bench.py --batch 256                     (1 GPU)
  python -m torch.distributed.run --nproc-per-node 8 ... scripts/dag_batch_bench.py --batch 256
Prints DAGs/s (whole job) on rank 0."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--logn", type=int, default=14)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--chunk", type=int, default=32, help="instances per batched device handle (0: one execute() per instance)")
args = ap.parse_args()

from eva_amd.dist import Dist
d = Dist()
from eva.ckks import CKKSCompiler
from eva.seal import generate_keys
from eva_amd.workloads import sobel as _sobel

prog = _sobel(64, 64, 4096); prog.set_input_scales(25); prog.set_output_ranges(10)
compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(prog)
params.poly_modulus_degree = 1 << args.logn
pub, sec = generate_keys(params, 1)          # every rank derives the same keys from the seed
pub.device = d.local_rank
units = d.my_units(args.batch)
encs = {u: pub.encrypt({'image': [((37 * i + u) % 256) / 255.0 for i in range(4096)]}, sig) for u in units[:8]}
keys = list(encs)
for u in keys[:2]:
    pub.execute(compiled, encs[u])           # eager walk, then graph capture
best = None
if args.chunk:
    pub.batch_chunk = args.chunk
    pub.execute_batch(compiled, [encs[keys[i % len(keys)]] for i in range(min(len(units), args.chunk))])  # warm-up
for _ in range(args.reps):
    def body():
        if args.chunk:
            pub.execute_batch(compiled, [encs[keys[i % len(keys)]] for i in range(len(units))])
            return
        for i, u in enumerate(units):
            pub.execute(compiled, encs[keys[i % len(keys)]])
        pub.synchronize()
    _, secs = d.timed(body)
    best = secs if best is None else min(best, secs)
if d.rank == 0:
    print(json.dumps({"workload": f"{args.batch} independent Sobel DAGs, N=2^{args.logn}, primes={list(params.prime_bits)}",
                      "n_gpus": d.world, "instances_per_handle": args.chunk, "dags_per_s": round(args.batch / best, 1), "ms_per_dag_per_gpu": round(best * 1e3 / len(units), 3)}))
d.close()
