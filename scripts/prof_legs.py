"""One bench leg under a profiler: python scripts/prof_legs.py batch|harris|execute [reps]
(rocprofv3 --kernel-trace --stats -- python scripts/prof_legs.py batch 2)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (the HIP runtime torch bundles is the one every library shares)
import bench  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "batch"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if which == "batch":
    print(json.dumps(bench.dag_batch_leg(256, reps)))
elif which == "harris":
    print(json.dumps(bench.dag_leg(reps, 8)))
else:
    print(json.dumps(bench.execute_leg(1 << 16, 10, 32, reps)))
