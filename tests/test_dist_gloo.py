"""N>1 path on CPU: two gloo ranks shard independent DAG instances (the unit the execute() path
shards over), run the host-side reference semantics on their share, and rank 0 reassembles the
batch — the same Dist helpers bench.py and scripts/dag_batch_bench.py use with RCCL on GPUs."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
    from eva_amd.dist import Dist, run_sharded
    from eva import evaluate
    from eva.ckks import CKKSCompiler
    from eva_amd.workloads import sobel as _sobel
    d = Dist(backend="gloo")
    assert d.world == 2 and d.backend == "gloo"
    prog = _sobel(8, 8, 64); prog.set_input_scales(25); prog.set_output_ranges(10)
    compiled, params, sig = CKKSCompiler(config={"warn_vec_size": "false"}).compile(prog)
    n_units = 7
    def work(u):
        img = [((37 * i + 11 * u) %% 256) / 255.0 for i in range(64)]
        return evaluate(compiled, {"image": img})["image"][:4]
    results, secs = run_sharded(d, n_units, work)
    mine = d.my_units(n_units)
    total = d.sum_over_ranks(len(mine))
    worst = d.max_over_ranks(d.rank + 1.5)
    if d.rank == 0:
        print("RESULT " + json.dumps({"results": results, "secs": secs, "total": total, "worst": worst, "mine": mine}))
    d.close()
""") % (ROOT, ROOT)


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][0]
    r = json.loads(line[len("RESULT "):])
    assert r["mine"] == [0, 2, 4, 6] and r["total"] == 7 and r["worst"] == 2.5 and r["secs"] > 0
    # serial reference
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from eva import evaluate
    from eva.ckks import CKKSCompiler
    from eva_amd.workloads import sobel as _sobel
    prog = _sobel(8, 8, 64)
    prog.set_input_scales(25)
    prog.set_output_ranges(10)
    compiled, _, _ = CKKSCompiler(config={"warn_vec_size": "false"}).compile(prog)
    for u in range(7):
        img = [((37 * i + 11 * u) % 256) / 255.0 for i in range(64)]
        assert r["results"][u] == evaluate(compiled, {"image": img})["image"][:4]


def test_unit_assignment_is_a_partition():
    from eva_amd.dist import Dist
    for world in (1, 2, 4, 8):
        seen = []
        for rank in range(world):
            d = Dist.__new__(Dist)
            d.rank, d.world = rank, world
            seen += d.my_units(256)
        assert sorted(seen) == list(range(256))
