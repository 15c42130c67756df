"""bench.py's multi-rank branches on a 1-GPU box: `--gpus 2` with the collectives around the data path over gloo
(EVA_BENCH_BACKEND=gloo) and both ranks on the one GPU (eva_amd/dist.py: rank r on device r mod count).  Everything a
multi-GPU driver run executes — the self-launch through torch.distributed.run, the per-rank key pairs and valuations,
the barrier-bracketed timing with max over ranks, the rank-sharded config-4 leg (instance b on rank b mod world,
SURVEY.md 8(e) row 1), `--shard dag`, `--shard limb` through the C++ limb-shard evaluator with its exchange hooks
(8(e) row 3) and `--shard subdag` — runs here first, each line checked against the oracle by bench.py itself (a
mismatch withholds the line and fails the command).  What stays hardware-only is RCCL itself (two ranks cannot
share a GPU under RCCL) and xGMI; DESIGN.md section 6 lists it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, timeout=420):
    env = dict(os.environ, EVA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, f"bench.py {' '.join(flags)} failed:\n{p.stdout[-2000:]}\n{p.stderr[-4000:]}"
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line expected on rank 0, got {len(lines)}:\n{p.stdout[-2000:]}"
    return json.loads(lines[0])


def test_two_ranks_default_command():
    """the driver's multi-GPU command at N = 2: op-triples through execute() on two ranks + the rank-sharded Sobel batch"""
    j = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--dag-batch", "64", "--no-cpu-baseline")
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["metric"].startswith("homomorphic ops/sec (mul+rescale+relin) at N=2^16, L=10")
    cfg = j["config"]
    assert cfg["entry_point"] == "public_ctx.execute"
    assert cfg["ranks"] == 2 and cfg["collectives_backend"] == "gloo" and cfg["rccl_ranks"] == 0  # gloo ranks are not RCCL ranks
    assert len(cfg["rank_devices"]) == 2 and cfg["rank_devices"][0].startswith("rank 0") and cfg["rank_devices"][1].startswith("rank 1")
    assert cfg["ciphertexts_moved_over_pcie_in_timed_region"] == {"ct_uploads": 0, "ct_downloads": 0}
    assert j["verified"]["bit_exact_vs_oracle"] is True and j["verified"]["triples_checked"] >= 2
    # value = the triples of both ranks over the max time: two ranks of 64 triples x 2 steps
    assert abs(j["value"] - 2 * 64 * 2 / (j["ms_per_step"] * 2e-3)) / j["value"] < 0.02
    leg = j["dag_batch"]
    assert "error" not in leg, leg
    assert leg["ranks"] == 2 and leg["instances_per_rank"] == 32 and leg["bit_exact_vs_oracle"] is True
    assert "rank b mod world" in leg["partition"]
    assert j["roofline"]["bound"] == "valu" and 0 < j["roofline"]["frac"] < 1


def test_two_ranks_shard_dag():
    j = _bench("--gpus", "2", "--shard", "dag", "--steps", "3", "--dag-batch", "64")
    assert j["n_gpus"] == 2 and j["unit"] == "DAGs/s" and j["scaling"] == "strong"
    assert j["config"]["ranks"] == 2 and len(j["config"]["rank_devices"]) == 2
    assert j["verified"]["bit_exact_vs_oracle"] is True and j["dag_batch"]["instances_per_rank"] == 32


def test_two_ranks_shard_limb():
    """every op-triple computed by both ranks together: the C++ limb-shard evaluator behind execute(), exchange hooks over gloo"""
    j = _bench("--gpus", "2", "--shard", "limb", "--steps", "2", "--warmup", "1", "--batch", "4")
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["unit"] == "op-triples/s"
    cfg = j["config"]
    assert cfg["entry_point"].startswith("public_ctx.execute") and cfg["ranks"] == 2 and cfg["collectives_backend"] == "gloo"
    assert cfg["exchange_launches_per_execute"] > 0 and cfg["exchanged_words_per_execute"] > 0
    assert j["verified"]["bit_exact_vs_oracle"] is True


def test_one_process_limb_shards_on_one_gpu():
    j = _bench("--gpus", "1", "--shard", "limb", "--shards", "4", "--steps", "2", "--warmup", "1", "--batch", "4")
    assert j["n_gpus"] == 1 and j["verified"]["bit_exact_vs_oracle"] is True
    kb = j["config"]["key_bytes"]
    assert len(kb["per_shard_here"]) == 4 and all(0 < b < kb["whole_key"] for b in kb["per_shard_here"])


def test_two_ranks_shard_subdag():
    j = _bench("--gpus", "2", "--shard", "subdag", "--steps", "2", "--warmup", "2")
    assert j["n_gpus"] == 2 and j["verified"]["bit_exact_vs_oracle"] is True
    assert len(j["config"]["members"]) == 2 and len(j["config"]["plan"]) >= 3
