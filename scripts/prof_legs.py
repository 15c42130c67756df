"""One bench leg under a profiler: python scripts/prof_legs.py batch|harris|harris_batch|c1|c2|c5|execute [reps]
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
elif which == "harris_batch":
    chunk = int(sys.argv[3]) if len(sys.argv) > 3 else None
    print(json.dumps(bench.dag_batch_leg(64, reps, workload="harris", chunk=chunk, check=2)))
elif which in ("c1", "c2", "c5"):
    # resident replays only (what the traces are read for): eager walk, capture, then `reps` replays
    from eva.seal import generate_keys
    from eva_amd import workloads
    compiled, params, sig, inputs = workloads.compile_config(which)
    pub, sec = generate_keys(params, 1)
    enc = pub.encrypt(inputs, sig)
    for _ in range(2 + reps):
        out = pub.execute(compiled, enc)
        pub.synchronize()
    print(json.dumps({"config": which, "replays": reps}))
elif which.startswith("pmc_"):
    # counter passes: exactly `reps` executions (or batch calls) and nothing else but the set-up encryption, so that the
    # sums of a FETCH_SIZE / WRITE_SIZE pass divide by a known number of DAGs (printed as "executions")
    leg = which[4:]
    if leg in ("c1", "c2", "c3", "c5"):
        from eva.seal import generate_keys
        from eva_amd import workloads
        compiled, params, sig, inputs = workloads.compile_config(leg)
        pub, sec = generate_keys(params, 1)
        enc = pub.encrypt(inputs, sig)
        for _ in range(reps):
            out = pub.execute(compiled, enc)
            pub.synchronize()
        print(json.dumps({"leg": leg, "executions": reps}))
    else:
        workload, batch = ("harris", 64) if leg == "harris_batch" else ("sobel", 256)
        state = bench._dag_batch_setup(batch, 0, 1, 0, 1, workload, None)  # (three warm-up calls inside)
        pub, compiled, inputs = state[0], state[2], state[5]
        for _ in range(reps):
            outs = pub.execute_batch(compiled, inputs)
            outs = None
        print(json.dumps({"leg": leg, "executions": (3 + reps) * batch}))
else:
    print(json.dumps(bench.execute_leg(1 << 16, 10, 32, reps)))
