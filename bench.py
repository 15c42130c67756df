#!/usr/bin/env python
"""bench.py — BASELINE.json's metric on MI355X: homomorphic op-triples/s
(multiply + relinearize + rescale) at N = 2^16, L = 10 data limbs (k = 11 key primes).

One "step" = one batch of --batch independent op-triples through the C-ABI of libeva_hip.so
(per group of --group triples: evah_multiply_many, then evah_relinearize_rescale_many =
relinearize and rescale_to_next evaluated together, bit-identical to the separate calls), inputs and the relinearization key already
resident in HBM.  One process per GPU; ranks run independent batches (the path shards over
independent ciphertexts — no data-path collective), `value` = triples of all ranks / max time.

Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline     — dominant kernel class by HIP-event time measured live in the timed region
  cpu_baseline — the CPU oracle (kind "port") timed on one host core on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def class_bytes(N, l, k, G=1):
    """Compulsory HBM bytes (distinct inputs read once + outputs written once) per launch of each
    kernel class for a group of G op-triples as bench.py issues it — evah_multiply_many, then
    evah_relinearize_rescale_many (relinearize at l limbs fused with the rescale l -> l-1);
    DESIGN.md §4.  Every launch covers the G triples of the group; the relinearization key is one
    input of the fused key-switch launch however many triples share it, so it is counted once
    per launch (the op-level figure below keeps SURVEY §8(d)'s per-op key bytes).
    W = one limb of one polynomial = 8N bytes."""
    W = 8 * N
    per_triple = {
        "elementwise": [7 * l * W],                                   # multiply: 4 polys in, 3 out
        "intt_pass1": [2 * l * W, 2 * 2 * W, (2 + 2 + 2 + 2) * W],   # digits; special limbs; t_K (a,prod,r in / t out)
        "intt_pass2": [2 * l * W, 2 * 2 * W, (2 + 2 + 2 + 2) * W],
        "ksdigit_pass1": [(l + l * l) * W],                           # l digits in, l^2 converted digits out
        "ks_mac": [(l * l + l + 2 * (l + 1)) * W],                    # digits + target in, prod out (key: below)
        "moddown_pass1": [(2 + 2 + 2 * (l - 1)) * W],                 # r, t in; intermediates out
        "moddown_pass2": [(4 * 2 * (l - 1)) * W],                     # interm + a + prod in; out
    }
    out = {kk: (sum(v) / len(v)) * G for kk, v in per_triple.items()}
    out["ks_mac"] += 2 * l * (l + 1) * W                              # the shared key, read once per launch
    return out


def triple_bytes(N, l):
    """multiply 7P + relinearize 5P + 2l(l+1)N8 + rescale 2P + 2(l-1)N8."""
    P = l * N * 8
    return 7 * P + 5 * P + 2 * l * (l + 1) * N * 8 + 2 * P + 2 * (l - 1) * N * 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="independent op-triples per step")
    ap.add_argument("--logn", type=int, default=16)
    ap.add_argument("--limbs", type=int, default=10)
    ap.add_argument("--streams", type=int, default=1,
                    help="issue queues (forked contexts = HIP streams) the independent triples are spread over")
    ap.add_argument("--group", type=int, default=32,
                    help="triples handed to one evah_relinearize_rescale_many call (wide launches, shared key)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import numpy as np
    # torch first (inside Dist): the HIP runtime torch bundles must be the one every library shares
    from eva_amd.dist import Dist
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); "
                         "the product path has no CPU fallback")
    dist = Dist(backend="nccl")
    rank, world, local = dist.rank, dist.world, dist.local_rank
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    from eva_amd import backend
    from eva_amd.hostref import coeff_modulus_create

    N, l = 1 << args.logn, args.limbs
    k = l + 1
    primes = coeff_modulus_create(N, [60] * k)
    g = backend.Context(N, primes, device=local)
    queues = [g] + [g.fork() for _ in range(max(1, args.streams) - 1)]

    # synthetic inputs: uniform residues; every triple of a step has its own
    # operand pair (distinct HBM data: no triple finds its inputs in cache because another used them)
    rng = np.random.default_rng(0xE7A + rank)

    def rand(prefix, nl):
        return np.stack([rng.integers(0, primes[i], size=prefix + (N,), dtype=np.uint64)
                         for i in range(nl)], axis=len(prefix))

    key_host = rand((l, 2), k)
    g.upload_relin_key(key_host)
    npairs = args.batch
    host_pairs, pairs = [], []
    for i in range(npairs):
        a, b = rand((2,), l), rand((2,), l)
        pairs.append((g.upload_ct(a, 2.0 ** 40), g.upload_ct(b, 2.0 ** 40)))
        if i < 4:
            host_pairs.append((a, b))  # the CPU baseline leg runs the first four

    PROF_EVERY = 8  # HIP-event brackets on every 8th triple only: keeps the timed region honest

    G = max(1, min(args.group, 64, args.batch))

    def step(profile=False):
        # the batch is issued in groups of G independent triples per queue: G multiplies, then one
        # relinearize_rescale_many (== rescale(relinearize(m)) for each m) as one wide launch set
        for gi, i0 in enumerate(range(0, args.batch, G)):
            q = queues[gi % len(queues)]
            sample = profile and (gi % max(1, PROF_EVERY // G) == 0)
            if sample:
                q.profile(True)
            idx = range(i0, min(i0 + G, args.batch))
            if len(idx) > 1:
                ms = q.multiply_many([pairs[i % npairs][0] for i in idx], [pairs[i % npairs][1] for i in idx])
                outs = q.relinearize_rescale_many(ms, 60)
            else:
                ms = [q.multiply(*pairs[i0 % npairs])]
                outs = [q.relinearize_rescale(ms[0], 60)]
            if sample:
                q.profile(False)
            for h in ms + outs:
                h.free()

    def barrier():
        for q in queues:
            q.sync()
        dist.barrier()  # torch.cuda.synchronize() + RCCL barrier

    for _ in range(args.warmup):
        step()
    barrier()
    for q in queues:
        q.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(profile=True)
    barrier()
    dt = time.perf_counter() - t0
    dt = dist.max_over_ranks(dt)
    prof = {}
    for q in queues:
        for c, (n_, ms_) in q.profile_get().items():
            a_ = prof.get(c, (0, 0.0))
            prof[c] = (a_[0] + n_, a_[1] + ms_)

    triples = args.steps * args.batch * world
    value = triples / dt

    # after the timed region: the same groups on ONE queue with nothing else in flight, so each
    # kernel has the chip to itself (the timed region overlaps two queues, which stretches every
    # launch it brackets); reported beside the timed-region figures, never instead of them
    iso = {}
    q0 = queues[0]
    q0.profile_reset()
    q0.profile(True)
    for _ in range(3):
        if G > 1:
            ms = q0.multiply_many([pairs[i % npairs][0] for i in range(G)], [pairs[i % npairs][1] for i in range(G)])
            outs = q0.relinearize_rescale_many(ms, 60)
        else:
            ms = [q0.multiply(*pairs[0])]
            outs = [q0.relinearize_rescale(ms[0], 60)]
        for h in ms + outs:
            h.free()
    q0.profile(False)
    q0.sync()
    iso = q0.profile_get()

    if rank == 0:
        cb = class_bytes(N, l, k, G)
        dom = max(prof, key=lambda c: prof[c][1])
        n_l, ms = prof[dom]
        avg_us = ms * 1e3 / max(n_l, 1)
        ach = cb[dom] / (avg_us * 1e-6) / 1e9 if dom in cb and n_l else 0.0
        kern_total_ms = sum(v[1] for v in prof.values())
        traffic = None  # PMC bytes per launch of the dominant kernel, from the committed rocprofv3 passes
        tpath = os.path.join(ROOT, "profiles", "bench_pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath))["by_class"].get(dom, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {
            "bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": traffic,
            "avg_launch_us": round(avg_us, 2), "launches_sampled": n_l,
            "sampling": (f"HIP events around every launch of 1 in {max(1, PROF_EVERY // G)} groups of {G} triples "
                         "inside the timed region"),
            "bytes_per_launch": int(cb.get(dom, 0)),
            "op_level": {  # SURVEY.md §8(d) figure: 198.2 MB per op-triple at N=2^16, l=10
                "bytes_per_triple": triple_bytes(N, l),
                "achieved": round(triple_bytes(N, l) * value / world / 1e9, 1),
                "frac": round(triple_bytes(N, l) * value / world / 1e9 / HBM_PEAK_GBPS, 4)},
            "isolated": {  # same launches, one queue, nothing overlapping (outside the timed region)
                "kernel": dom,
                "avg_launch_us": round(iso[dom][1] * 1e3 / max(iso[dom][0], 1), 2) if dom in iso else None,
                "achieved": round(cb[dom] / (iso[dom][1] * 1e-3 / max(iso[dom][0], 1)) / 1e9, 1)
                if dom in iso and iso[dom][1] > 0 and dom in cb else None,
                "by_class_us": {c: round(v[1] * 1e3 / max(v[0], 1), 2) for c, v in iso.items() if v[0]}},
            "by_class_us": {c: round(v[1] * 1e3 / max(v[0], 1), 2) for c, v in prof.items() if v[0]},
            "by_class_share": {c: round(v[1] / kern_total_ms, 3) for c, v in prof.items() if v[0]},
        }
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # reported baseline: rank 0 at N=1 only
            from oracle import pyoracle as po  # checker / reported baseline only
            o = po.Oracle(N, primes)
            a, b = host_pairs[0]
            t1 = time.perf_counter()
            o.op_triple(a, b, key_host)
            one = time.perf_counter() - t1
            n = max(1, min(50, int(args.cpu_seconds / max(one, 1e-3))))
            t1 = time.perf_counter()
            for i in range(n):
                a, b = host_pairs[i % len(host_pairs)]
                o.op_triple(a, b, key_host)
            cdt = time.perf_counter() - t1
            # the same port on many host cores at once (independent triples, one per thread; ctypes
            # releases the GIL) — the analogue of the reference's Galois node-level parallelism
            import threading
            threads = max(1, min(os.cpu_count() or 1, 64))
            done = []

            def worker(i):
                a_, b_ = host_pairs[i % len(host_pairs)]
                o.op_triple(a_, b_, key_host)
                done.append(i)
            ths = [threading.Thread(target=worker, args=(i,)) for i in range(threads)]
            t1 = time.perf_counter()
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            mdt = time.perf_counter() - t1
            cpu = {"value": round(n / cdt, 3), "unit": "op-triples/s", "cores": 1, "kind": "port",
                   "all_cores": {"value": round(len(done) / mdt, 2), "cores": threads,
                                 "sample": f"{threads} op-triples, one per thread, concurrently"},
                   "sample": f"{n} op-triples (multiply+relinearize+rescale) at N=2^{args.logn}, "
                             f"L={l}, same inputs/key as the GPU run, oracle/libeva_oracle.so, "
                             f"1 thread of {os.cpu_count()} host cores"}
        line = {
            "metric": "homomorphic ops/sec (mul+rescale+relin) at N=2^16, L=10; execute() wall-time",
            "value": round(value, 2), "unit": "op-triples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt * 1e3 / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"op-triple multiply+relinearize+rescale, N=2^{args.logn}, "
                                   f"L={l} data limbs + 1 special prime (60-bit), "
                                   f"{args.batch} independent triples per step per GPU",
                       "poly_modulus_degree": N, "limbs": l, "batch_per_gpu": args.batch,
                       "streams_per_gpu": len(queues), "triples_per_call": G,
                       "parallelism": f"independent ciphertexts sharded over {world} GPU(s), no collective"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    # orderly teardown: values, then forked queues, then the root context
    for a, b in pairs:
        a.free(); b.free()
    for q in queues[1:]:
        q.close()
    g.close()
    dist.close()


if __name__ == "__main__":
    main()
