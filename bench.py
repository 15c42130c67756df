#!/usr/bin/env python
"""bench.py — BASELINE.json's metric on MI355X: homomorphic op-triples/s
(multiply + relinearize + rescale) at N = 2^16, L = 10 data limbs (k = 11 key primes), **execute() wall-time**.

One "step" = --batch independent op-triples through `public_ctx.execute()` — the reference's
SEALPublic::execute (/root/reference/eva/seal/seal.cpp:104-122, python binding wrapper.cpp:215): a compiled EVA
program of --group products z_i = x_i * y_i (Mul -> Relinearize -> Rescale each, eager relinearization), one
execute() call per group on that group's own encrypted valuation.  The valuations were left in HBM by encrypt()
(a valuation "may hold device handles"), so inputs are resident when the timed region starts, the
outputs stay resident, execute() is an asynchronous enqueue and consecutive calls alternate between the context's
two issue queues.  Behind execute() the library scheduler runs each group as ONE
evah_multiply_relinearize_rescale_many (bit-identical to the three separate SEAL calls).  One process per GPU;
ranks run independent batches (the path shards over independent ciphertexts — no data-path collective), `value` =
triples of all ranks / max time over ranks.

`python bench.py --gpus N` with no torch.distributed environment launches the N ranks itself
(python -m torch.distributed.run, 127.0.0.1); under torchrun it is one rank of the job.  EVA_BENCH_BACKEND=gloo
runs the collectives around the data path over gloo, ranks sharing the visible GPUs (rank r on device r mod
count) — how a 1-GPU box exercises every multi-rank branch of this file (tests/test_gpu_bench_ranks.py).

Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline     — SURVEY.md section 8(d) algorithmic bytes of the op-triple x value over the 8 TB/s HBM peak
                 (`frac` = `hbm_frac`), the VALU-issue fraction from the committed SQ-counter pass (`valu_frac`; the
                 path is integer-issue bound, `bound` says so), the compulsory bytes of one launch set with the key
                 counted once, and the dominant kernel class by HIP-event time measured live inside the timed region
                 on the launch streams
  verified     — after the timed region one output per group is downloaded and compared, word for
                 word, with the CPU oracle's multiply+relinearize+rescale of the same operands; the same outputs are
                 decrypted and compared with x * y
  raw_cabi     — the same triples by direct C-ABI calls (evah_multiply_relinearize_rescale_many through ctypes),
                 reported beside the headline, never as it
  execute_path — host-valuation variants of the headline (pipelined host inputs, full host round trip: the
                 PCIe-inclusive figures, never `value`)
  dag          — Harris corner detector, N = 2^15, L = 8 (BASELINE config 3): execute() ms and the
                 CPU walk of the same compiled DAG over the oracle (serial and all host cores),
                 north_star's ">= 10x the CPU on Harris" driver-timed
  cpu_baseline — the CPU oracle (kind "port"; "SEAL absent" unless a real SEAL is installed on the
                 host or a pin-kit result is present, tools/pin_with_seal.sh) on one host core on a bounded sample,
                 and on many cores
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def class_bytes(N, l, k, G=1):
    """Distinct HBM bytes (inputs read once + outputs written once) per launch of each kernel class
    for a group of G op-triples; the relinearization key is one input of the fused key-switch
    launch however many triples share it.  W = one limb of one polynomial = 8N bytes.  These are
    per-KERNEL figures (they include the l^2 converted digits that SURVEY 8(d) counts as NTT-internal)
    and only feed `roofline.dominant`; the headline uses triple_bytes()."""
    W = 8 * N
    per_triple = {
        "elementwise": 7 * l * W,
        "intt_pass1": (2 * l * W + 2 * 2 * W + 8 * W) / 3, "intt_pass2": (2 * l * W + 2 * 2 * W + 8 * W) / 3,
        "ksdigit_pass1": (l + l * l) * W,
        "ks_mac": (l * l + l + 2 * (l + 1)) * W,
        "moddown_pass1": (2 + 2 + 2 * (l - 1)) * W,
        "moddown_pass2": (4 * 2 * (l - 1)) * W,
    }
    out = {kk: v * G for kk, v in per_triple.items()}
    out["ks_mac"] += 2 * l * (l + 1) * W
    return out


def triple_bytes(N, l):
    """multiply 7P + relinearize 5P + 2l(l+1)N8 + rescale 2P + 2(l-1)N8."""
    P = l * N * 8
    return 7 * P + 5 * P + 2 * l * (l + 1) * N * 8 + 2 * P + 2 * (l - 1) * N * 8


def self_launch(args):
    """--gpus N without a torch.distributed environment: start the N ranks here."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    shared = os.environ.get("EVA_BENCH_BACKEND", "nccl") != "nccl"  # gloo: ranks share the visible GPUs (rank r on device r mod count)
    if have < (1 if shared else args.gpus):
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this node")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def pad_chain(params, n_primes, N):
    """pad the prime chain to the stated L (eva_amd/workloads.py; imported late — torch must load its HIP
    runtime before the library does)"""
    from eva_amd.workloads import pad_chain as f
    return f(params, n_primes, N)


def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def host_cores():
    """Cores this process may run on — how the reference sizes its thread pool (python/eva/__init__.py:10-14)."""
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return max(1, os.cpu_count() or 1)


def triple_program(n_products, N, l):
    """z_i = x_i * y_i for i < n_products, compiled with eager relinearization: every product is
    Mul -> Relinearize -> Rescale on l data limbs (k = l + 1 key primes, all 60-bit) -> (compiled, params, signature)"""
    from eva import EvaProgram, Input, Output
    from eva.ckks import CKKSCompiler
    prog = EvaProgram('op_triples', vec_size=1024)
    with prog:
        for i in range(n_products):
            Output(f'z{i}', Input(f'x{i}') * Input(f'y{i}'))
    prog.set_input_scales(60)
    prog.set_output_ranges(20)
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false', 'lazy_relinearize': 'false'}).compile(prog)
    params.poly_modulus_degree = N
    params.prime_bits = [60] * (l + 1)
    return compiled, params, sig


def host_valuation_legs(pub, compiled, enc, n_products, l, N, reps, want):
    """The headline's execute() with the valuations held on the HOST instead of in HBM (the PCIe-inclusive figures,
    never `value`):
      pipelined_host  host inputs (pinned), resident outputs: each call blocks in its own uploads, which
                      overlap the previous call's kernels on the other issue queue
      host_roundtrip  host valuations in and out (EVA_RESIDENT=0 behaviour): upload, run, download, per call
    `want`: z0 of the resident run (the host paths must give the same words)."""
    import numpy as np
    enc.to_host(True)                           # host words only from here on (pinned pages)

    def timed(n):
        for _ in range(2):
            out = pub.execute(compiled, enc)
        pub.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = pub.execute(compiled, enc)
        pub.synchronize()
        return (time.perf_counter() - t0) / n, out
    t_pipe, out2 = timed(max(3, reps // 2))
    ok = bool(np.array_equal(out2.get('z0')[4], want))
    pub.resident = False
    ts, parts = [], []
    for _ in range(max(3, reps // 2) + 2):
        t0 = time.perf_counter()
        out3 = pub.execute(compiled, enc)
        ts.append(time.perf_counter() - t0)
        parts.append(list(pub.last_timing))
    pub.resident = True
    ts, parts = ts[2:], parts[2:]
    t_host = _median(ts)
    ok = ok and bool(np.array_equal(out3.get('z0')[4], want))
    pm = [_median([p_[j] for p_ in parts]) for j in range(3)]
    in_mb = 2 * n_products * 2 * l * N * 8 / 1e6
    return {"pipelined_host_inputs": {"triples_per_s": round(n_products / t_pipe, 1), "ms_per_execute": round(t_pipe * 1e3, 3),
                                      "input_mb_per_execute": round(in_mb, 1), "pcie_gb_per_s": round(in_mb / 1e3 / t_pipe, 1),
                                      "note": "PCIe-bound: 21 MB of operands per triple; uploads overlap the previous call's kernels"},
            "host_roundtrip": {"triples_per_s": round(n_products / t_host, 1), "ms_per_execute": round(t_host * 1e3, 3),
                               "ms_upload_enqueue_drain": [round(x, 3) for x in pm],
                               "includes": "input upload (PCIe), run, output download (PCIe), synchronous"},
            "same_words_as_resident": ok}


def raw_cabi_leg(args, N, l, primes, device, steps, warmup):
    """The same op-triples by direct C-ABI calls through ctypes (one evah_multiply_relinearize_rescale_many per group,
    groups alternating between --streams issue queues, uniform random residues and key): what the library does without
    the host side of execute() around it.  Reported beside the headline, never as it."""
    import numpy as np
    from eva_amd import backend
    k = l + 1
    g = backend.Context(N, primes, device=device)
    queues = [g] + [g.fork() for _ in range(max(1, args.streams) - 1)]
    rng = np.random.default_rng(0xE7A)

    def rand(prefix, nl):
        return np.stack([rng.integers(0, primes[i], size=prefix + (N,), dtype=np.uint64)
                         for i in range(nl)], axis=len(prefix))
    key_host = rand((l, 2), k)
    g.upload_relin_key(key_host)
    G = max(1, min(args.group, 64, args.batch))
    pairs, host0 = [], None
    for i in range(args.batch):
        a, b = rand((2,), l), rand((2,), l)
        pairs.append((g.upload_ct(a, 2.0 ** 40), g.upload_ct(b, 2.0 ** 40)))
        if i == 0:
            host0 = (a, b)

    def step(keep=None):
        for gi, i0 in enumerate(range(0, args.batch, G)):
            q = queues[gi % len(queues)]
            idx = range(i0, min(i0 + G, args.batch))
            As, Bs = [pairs[i][0] for i in idx], [pairs[i][1] for i in idx]
            if args.separate_multiply:
                ms = q.multiply_many(As, Bs)
                outs = q.relinearize_rescale_many(ms, 60)
                hs = ms + outs
            else:
                outs = q.multiply_relinearize_rescale_many(As, Bs, 60)
                hs = outs
            if keep is not None and i0 == 0:
                keep.append(outs[0].download())
            for h in hs:
                h.free()

    def sync():
        for q in queues:
            q.sync()
    for _ in range(warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    got = []
    step(keep=got)
    sync()
    from oracle import pyoracle as po  # checker only
    ok = bool(np.array_equal(got[0], po.Oracle(N, primes).op_triple(host0[0], host0[1], key_host)))
    for a, b in pairs:
        a.free(); b.free()
    for q in queues[1:]:
        q.close()
    g.close()
    return {"triples_per_s": round(steps * args.batch / dt, 1), "ms_per_step": round(dt * 1e3 / steps, 4), "steps": steps,
            "entry_point": "evah_multiply_relinearize_rescale_many" if not args.separate_multiply else
                           "evah_multiply_many + evah_relinearize_rescale_many",
            "streams": len(queues), "triples_per_call": G, "inputs": "uniform random residues and key (SURVEY.md 8(d))",
            "bit_exact_vs_oracle": ok}


def dag_leg(reps, cpu_threads):
    """BASELINE config 3 / north_star's target: Harris corner detector at N = 2^15, L = 8."""
    import numpy as np
    from eva.ckks import CKKSCompiler
    from eva.seal import generate_keys
    from eva_amd.roofline import dag_bytes, dag_compulsory_bytes, roofline as rl
    from eva_amd.workloads import harris as _harris, image as _image
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(_harris())
    pad_chain(params, 9, 32768)
    pub, sec = generate_keys(params, 1)
    nbytes, by = dag_bytes(compiled, sig, 32768, 9)
    comp = dag_compulsory_bytes(compiled, sig, 32768, 9)
    enc = pub.encrypt(_image(4096), sig)
    # resident valuations: enqueue + synchronize per call (the latency of one execute())
    for _ in range(3):
        out = pub.execute(compiled, enc)
    pub.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = pub.execute(compiled, enc)
        pub.synchronize()
        ts.append(time.perf_counter() - t0)
    res_ms = _median(ts) * 1e3
    # the same replays back to back (no wait in between): the GPU's own time per replay, and the host's per call
    hs = []
    t0 = time.perf_counter()
    for _ in range(4 * reps):
        h0 = time.perf_counter()
        out = pub.execute(compiled, enc)
        hs.append(time.perf_counter() - h0)
    pub.synchronize()
    b2b_ms = (time.perf_counter() - t0) / (4 * reps) * 1e3
    host_us = _median(hs) * 1e6
    # host valuations: upload + replay + download, as the reference's execute() hands values over
    enc.to_host(True)
    pub.resident = False
    for _ in range(3):
        out_h = pub.execute(compiled, enc)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out_h = pub.execute(compiled, enc)
        ts.append(time.perf_counter() - t0)
    gpu_ms = _median(ts) * 1e3
    # CPU: the same compiled DAG walked in C over the oracle (checker / reported baseline only)
    from oracle.executor import c_walk
    ref, t1 = c_walk(pub, compiled, enc, threads=1)
    ok = all(np.array_equal(out_h.get(name)[4], ref[name]) and np.array_equal(out.get(name)[4], ref[name]) for name in ref)
    _, tn = c_walk(pub, compiled, enc, threads=cpu_threads)
    t64 = c_walk(pub, compiled, enc, threads=64)[1] if cpu_threads > 64 else None
    kinds = [str(d["op"]).split(".")[-1] for d in compiled._dump()]
    return {"workload": "Harris corner detector (examples/image_processing.py), 64x64 image, N=2^15, L=8 data limbs",
            "terms": len(kinds), "rotations": kinds.count("RotateLeftConst") + kinds.count("RotateRightConst"),
            "relinearize": kinds.count("Relinearize"), "rescale": kinds.count("Rescale"),
            "gpu_execute_ms": round(gpu_ms, 3), "gpu_includes": "input upload, hipGraph replay, output download (host valuations)",
            "gpu_execute_resident_ms": round(res_ms, 3),
            "resident_back_to_back_ms": round(b2b_ms, 3), "execute_returns_after_us": round(host_us, 1),
            "resident_note": "gpu_execute_resident_ms = execute() + synchronize() per call (latency); back to back = time per call "
                             "when the caller issues the next execute() while the previous replay runs (r6: the replays then "
                             "alternate between the program's twin plans, two chains side by side); execute() itself returns "
                             "to the host after execute_returns_after_us",
            "roofline": rl(nbytes, gpu_ms * 1e-3, compulsory=comp), "roofline_resident": rl(nbytes, res_ms * 1e-3, compulsory=comp),
            "cpu_walk_ms": dict({"1": round(t1 * 1e3, 1), str(cpu_threads): round(tn * 1e3, 1)},
                                **({"64": round(t64 * 1e3, 1)} if t64 else {})),
            "cpu_cores": cpu_threads,
            "cpu_walk": "oracle/eva_oracle_dag.c: serial forwardPass / dependency-counting traversal on pthreads",
            "speedup_vs_cpu": dict({"1": round(t1 * 1e3 / gpu_ms, 1), str(cpu_threads): round(tn * 1e3 / gpu_ms, 1)},
                                   **({"64": round(t64 * 1e3 / gpu_ms, 1)} if t64 else {})),
            "bit_exact_vs_oracle": bool(ok)}


BATCH_WORKLOADS = {
    # name -> (program builder, N, primes, what the line calls it, default instances per batched handle (None = the context's))
    "sobel": ("sobel_example", 16384, 6, "independent Sobel DAGs (examples/image_processing.py:39-63), 64x64 images", None),
    "harris": ("harris", 32768, 9, "independent Harris corner DAGs (examples/image_processing.py:65-100), 64x64 images", 12),
}


def dag_batch_leg(batch, reps, dist=None, members=1, workload="sobel", chunk=None, check=8):
    """A batch of independent DAGs of one program through execute_batch.  r6: the valuations are device-resident — encrypt()
    left the inputs in HBM, the outputs stay there (no ciphertext crosses PCIe in the timed calls: SURVEY.md 8(b), and the
    rule that inputs are resident when the timed region starts); the same call with host valuations (uploads and downloads
    included — what the leg timed through r5) is reported beside it as `host_valuations`.
      workload "sobel"   BASELINE config 4: Sobel at N = 2^14, L = 5
      workload "harris"  north_star's target as THROUGHPUT: Harris corner detector at N = 2^15, L = 8 (config 3's DAG)
    `check` instances per rank, spread over the groups and issue queues of the call, are compared word for word with
    the C walk of the oracle.  dist with world > 1: instance b runs on rank b mod world (one
    process per GPU, no data-path collective), `dags_per_s` = batch / max-over-ranks time.  members > 1 (one
    process): shard_mode = "dag" over `members` contexts of this rank's GPU."""
    world = dist.world if dist else 1
    rank = dist.rank if dist else 0
    dev = dist.device_index if dist else 0
    if dist and world > 1:
        # a rank that fails while the others wait in a barrier would stall the job: set up first, agree, then time
        err = None
        try:
            state = _dag_batch_setup(batch, rank, world, dev, members, workload, chunk)
        except Exception as e:  # noqa: BLE001
            err, state = repr(e), None
        if dist.sum_over_ranks(0.0 if err is None else 1.0) > 0:
            return {"error": err or "set-up failed on another rank"}
        return _dag_batch_run(state, batch, reps, dist, members, workload, check)
    return _dag_batch_run(_dag_batch_setup(batch, rank, world, dev, members, workload, chunk), batch, reps, dist, members, workload, check)


def _dag_batch_setup(batch, rank, world, dev, members, workload="sobel", chunk=None):
    from eva.ckks import CKKSCompiler
    from eva.seal import generate_keys
    from eva_amd import workloads
    from eva_amd.roofline import dag_bytes, dag_compulsory_bytes
    builder, N, n_primes, _, default_chunk = BATCH_WORKLOADS[workload]
    compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(getattr(workloads, builder)())
    pad_chain(params, n_primes, N)
    if members > 1:
        pub, sec = generate_keys(params, 1, devices=[dev] * members, shard="dag")
    elif dev:
        pub, sec = generate_keys(params, 1, devices=[dev])
    else:
        pub, sec = generate_keys(params, 1)
    pub.resident = members == 1  # (shard_mode "dag" deals host valuations over its members)
    if chunk or default_chunk:
        pub.batch_chunk = int(chunk or default_chunk)
    nbytes, _ = dag_bytes(compiled, sig, N, len(params.prime_bits))
    comp = dag_compulsory_bytes(compiled, sig, N, len(params.prime_bits), instances=int(pub.batch_chunk))
    encs = [pub.encrypt(workloads.image(4096, shift=u), sig) for u in range(8)]
    mine = list(range(rank, batch, world))  # instance b -> rank b mod world
    inputs = [encs[b % len(encs)] for b in mine]
    for _ in range(3):  # warm-up: tables, constants; the pools of the four issue queues reach their steady state in the second call
        pub.execute_batch(compiled, inputs)
    return pub, sec, compiled, params, (nbytes, comp), inputs, mine


def _spread(n, want):
    """`want` instance indices out of n: both ends and evenly in between, offset so they fall in different groups / queues"""
    if n <= want:
        return list(range(n))
    return sorted({0, n - 1} | {min(n - 1, (i * n) // (want - 1) + (i % 3)) for i in range(1, want - 1)})


def _dag_batch_run(state, batch, reps, dist, members, workload="sobel", check=8):
    import numpy as np
    from eva_amd.roofline import roofline as rl
    pub, sec, compiled, params, (nbytes, comp), inputs, mine = state
    world = dist.world if dist else 1
    ts, outs = [], None
    import gc
    gc.collect()
    gc.disable()  # (as timeit does: a generation-2 collection inside one call of ~17 ms is a visible outlier)
    for _ in range(reps):
        outs = None  # the previous call's output valuations go back to the pinned pool before the clock starts
        if dist:
            dist.barrier()
        t0 = time.perf_counter()
        outs = pub.execute_batch(compiled, inputs)
        if dist:
            dist.barrier()
        dt = time.perf_counter() - t0
        ts.append(dist.max_over_ranks(dt) if dist else dt)
    gc.enable()
    med = _median(ts)
    # the same call with host valuations: inputs as host words (pinned), outputs downloaded — PCIe included
    host_val = None
    if pub.resident and not dist:
        houts = None
        try:
            for e in {id(v): v for v in inputs}.values():
                e.to_host(True)
            pub.resident = False
            hts = []
            for _ in range(2 + max(3, reps // 2)):
                houts = None
                t0 = time.perf_counter()
                houts = pub.execute_batch(compiled, inputs)
                hts.append(time.perf_counter() - t0)
            hmed = _median(hts[2:])
            same = all(np.array_equal(houts[i].get(name)[4], outs[i].get(name)[4]) for i in (0, len(mine) - 1) for name in outs[i].names())
            host_val = {"dags_per_s": round(batch / hmed, 1), "ms_total": round(hmed * 1e3, 2), "same_words_as_resident": bool(same),
                        "includes": "input uploads and output downloads over PCIe (host valuations, as the reference hands values over)"}
        except Exception as e:  # noqa: BLE001
            host_val = {"error": repr(e)}
        finally:
            pub.resident = True
            houts = None
    from oracle.executor import c_walk  # checker only
    ok, picked, walked = True, _spread(len(mine), check), {}
    for i in picked:
        # (the batch cycles over 8 distinct encrypted images: one walk per distinct input, every picked instance compared)
        if id(inputs[i]) not in walked:
            walked[id(inputs[i])] = c_walk(pub, compiled, inputs[i], threads=min(host_cores(), 64))[0]
        ref = walked[id(inputs[i])]
        ok = ok and all(np.array_equal(outs[i].get(name)[4], ref[name]) for name in ref)
    bad = dist.sum_over_ranks(0.0 if ok else 1.0) if dist else (0.0 if ok else 1.0)
    chunk = int(pub.batch_chunk)
    sets = -(-len(mine) // chunk) + (2 if pub.batch_ramp and chunk >= 4 and len(mine) >= 4 * chunk else 0)  # (batch.h: ramped first / last groups)
    comp_rank = (comp[0] / chunk * len(mine) - comp[1]["key_bytes_once"] * (len(mine) / chunk - sets), dict(comp[1], launch_sets=sets))
    return {"workload": f"{batch} {BATCH_WORKLOADS[workload][3]}, N=2^{int(np.log2(params.poly_modulus_degree))}, primes={list(params.prime_bits)}",
            "dags_per_s": round(batch / med, 1), "ms_total": round(med * 1e3, 2), "best_dags_per_s": round(batch / min(ts), 1),
            "ms_calls": [round(t * 1e3, 2) for t in ts],
            "timing": f"median of {reps} calls" + (", barrier + max over ranks per call" if world > 1 else ""),
            "instances_per_device_handle": chunk, "instances_per_rank": len(mine), "ranks": world,
            "group_sizes": ("balanced: %d groups of %d or %d instances" % (sets, len(mine) // sets, -(-len(mine) // sets))
                            if getattr(pub, "batch_balance", False) and not pub.batch_ramp else "groups of %d and a remainder" % chunk),
            "members_per_rank": members, "shard_mode": "dag" if members > 1 else "",
            "partition": "instance b on rank b mod world (SURVEY.md 8(e) row 1), no data-path collective" if world > 1 else "one rank",
            "roofline": rl(nbytes * batch / world, med, compulsory=comp_rank),
            "roofline_basis": "algorithmic bytes of this rank's DAGs over the time, per GPU; launch_compulsory: every evaluation key "
                              f"charged once per group of {chunk} instances and scheduler level",
            "includes": (f"one DAG walk per {chunk} instances on device-resident valuations (inputs stacked device to device, outputs "
                         "left in HBM); no ciphertext crosses PCIe" if pub.resident else
                         f"input uploads, one DAG walk per {chunk} instances, output downloads"),
            "valuations": "device-resident" if pub.resident else "host", "host_valuations": host_val,
            "instances_checked_per_rank": len(picked), "instances_checked": [int(mine[i]) for i in picked],
            "bit_exact_vs_oracle": bad == 0.0}


def dag_configs_leg(reps):
    """"also report per-DAG execute() wall time for configs C1-C5".  C3 and C4 have legs of their own
    (`dag`, `dag_batch`); here C1 (README polynomial), C2 (Sobel, N = 2^13) and C5 (3x3 convolution + 8 squarings,
    N = 2^16, L = 12): execute() with host valuations (upload + run + download, as the reference hands values over) and
    with resident valuations (enqueue + synchronize), each against the DAG's algorithmic bytes, outputs compared word
    for word with the C walk of the oracle."""
    import numpy as np
    from eva.seal import generate_keys
    from eva_amd import workloads
    from eva_amd.roofline import dag_bytes, dag_compulsory_bytes, roofline as rl
    from oracle.executor import c_walk  # checker only
    out = {}
    for name in ("c1", "c2", "c5"):
        try:
            compiled, params, sig, inputs = workloads.compile_config(name)
            N, k = params.poly_modulus_degree, len(params.prime_bits)
            pub, sec = generate_keys(params, 1)
            enc = pub.encrypt(inputs, sig)
            nbytes, _ = dag_bytes(compiled, sig, N, k)
            comp = dag_compulsory_bytes(compiled, sig, N, k)
            for _ in range(3):  # eager walk, hipGraph capture, first replay
                res = pub.execute(compiled, enc)
            pub.synchronize()
            tr = []
            for _ in range(reps):
                t0 = time.perf_counter()
                res = pub.execute(compiled, enc)
                pub.synchronize()
                tr.append(time.perf_counter() - t0)
            # the same replays with nothing waiting in between: the caller issues while the previous one runs, the replays
            # alternate between the program's twin plans (public_ctx.h) — time per call with the queues kept full
            t0 = time.perf_counter()
            for _ in range(4 * reps):
                res = pub.execute(compiled, enc)
            pub.synchronize()
            b2b_s = (time.perf_counter() - t0) / (4 * reps)
            enc.to_host(True)
            pub.resident = False
            for _ in range(2):
                hout = pub.execute(compiled, enc)
            th = []
            for _ in range(reps):
                t0 = time.perf_counter()
                hout = pub.execute(compiled, enc)
                th.append(time.perf_counter() - t0)
            ref, t1 = c_walk(pub, compiled, enc, threads=1)
            _, tn = c_walk(pub, compiled, enc, threads=min(host_cores(), 64))
            ok = all(np.array_equal(hout.get(n)[4], ref[n]) and np.array_equal(res.get(n)[4], ref[n]) for n in ref)
            kinds = [str(d["op"]).split(".")[-1] for d in compiled._dump()]
            res_s, host_s = _median(tr), _median(th)
            out[name] = {"workload": {"c1": "README polynomial 3x^2+5x-2, vec_size 1024", "c2": "Sobel 3x3 filter, 64x64 image",
                                      "c5": "3x3 convolution + depth-8 squaring chain (tests/large_programs.py style)"}[name]
                                     + f", N=2^{int(np.log2(N))}, primes={list(params.prime_bits)}",
                         "terms": len(kinds), "rotations": kinds.count("RotateLeftConst") + kinds.count("RotateRightConst"),
                         "relinearize": kinds.count("Relinearize"), "rescale": kinds.count("Rescale"),
                         "host_ms": round(host_s * 1e3, 3), "resident_ms": round(res_s * 1e3, 3),
                         "resident_back_to_back_ms": round(b2b_s * 1e3, 3),
                         "roofline": rl(nbytes, host_s, compulsory=comp), "roofline_resident": rl(nbytes, res_s, compulsory=comp),
                         "cpu_walk_ms": {"1": round(t1 * 1e3, 1), str(min(host_cores(), 64)): round(tn * 1e3, 1)},
                         "bit_exact_vs_oracle": bool(ok)}
            res = hout = enc = pub = sec = None
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": repr(e)}
    return out


def dag_sharded(args, dist):
    """--shard dag: BASELINE config 4's scaling leg as the whole job — 256 independent Sobel DAGs at N = 2^14,
    l = 5, dealt b mod G over the ranks (one process per GPU, eva_amd/dist.py; --members k additionally splits a
    rank's groups over k contexts of its GPU with shard_mode = "dag").  One JSON line on rank 0: DAGs/s of the
    whole job and the achieved-HBM fraction per GPU."""
    import torch
    world, local = dist.world, dist.device_index
    dev_name = torch.cuda.get_device_name(local)
    rank_devices = [dev_name]
    if world > 1:
        import torch.distributed as tdist
        gathered = [None] * world
        tdist.all_gather_object(gathered, f"rank {dist.rank}: cuda:{local} {dev_name}")
        rank_devices = gathered
    leg = dag_batch_leg(args.dag_batch, max(3, args.steps), dist, members=max(1, args.members))
    if dist.rank == 0:
        if not leg["bit_exact_vs_oracle"]:
            raise SystemExit("bench.py --shard dag: an instance differs from the CPU oracle's walk — number withheld")
        line = {"metric": "independent Sobel DAGs/s (BASELINE config 4: batch of 256, N=2^14), execute_batch() wall-time",
                "value": leg["dags_per_s"], "unit": "DAGs/s", "n_gpus": world, "steps": max(3, args.steps), "warmup": 1,
                "ms_per_step": leg["ms_total"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "u64", "data": "synthetic",
                "config": {"workload": leg["workload"], "batch": args.dag_batch, "parallelism": leg["partition"],
                           "members_per_rank": leg["members_per_rank"], "ranks": world, "collectives_backend": dist.backend,
                           "rccl_ranks": world if dist.backend == "nccl" else 0, "rank_devices": rank_devices,
                           "collectives": "barrier + max-over-ranks of the wall time (torch.distributed "
                                          + ("nccl = RCCL" if dist.backend == "nccl" else dist.backend) + "); none in the data path"},
                "roofline": dict(leg["roofline"], bound="hbm", basis=leg["roofline_basis"]),
                "verified": {"instances_checked_per_rank": leg["instances_checked_per_rank"], "bit_exact_vs_oracle": True},
                "cpu_baseline": None, "dag_batch": leg}
        print(json.dumps(line), flush=True)
    dist.close()


def limb_sharded(args, dist):
    """--shard limb: every op-triple is computed by ALL GPUs together, the RNS limbs dealt over them (limb i on shard
    i mod G; SURVEY.md 8(e) row 3, BASELINE config 5's mode): per key switch one all-gather of the coefficient-form
    digits and one broadcast, per rescale one broadcast.  The timed region is public_ctx.execute() of the op-triple
    program with shard_mode = "limb" — the C++ limb-shard evaluator (eva_amd/host/multi_device.h) is the only driver
    of the protocol.  Under torchrun every rank is one shard and its exchange steps are collectives on the library's
    device buffers (eva_amd.dist.attach_limb_dist: RCCL, or gloo staged through the host under EVA_BENCH_BACKEND=gloo);
    with one process, --shards G runs G shards on the one GPU with one gather launch per receiving shard as the
    exchange (what a 1-GPU box can measure: the cost of the phase structure, not xGMI)."""
    import numpy as np
    import torch.distributed as tdd
    from eva.seal import generate_keys, SEALValuation
    from eva_amd.dist import attach_limb_dist
    N, l = 1 << args.logn, args.limbs
    world = dist.world
    n_products = max(1, min(args.batch, 8))
    compiled, params, sig = triple_program(n_products, N, l)
    G = world if world > 1 else max(2, args.shards)
    if world > 1:
        pub, sec = generate_keys(params, 17)  # the same keys on every rank (seeded)
        pub.device = dist.device_index
    else:
        pub, sec = generate_keys(params, 17, devices=[dist.device_index] * G, shard="limb")
    rng = np.random.default_rng(5)
    inputs = {}
    for i in range(n_products):
        inputs[f'x{i}'] = list(rng.uniform(-1, 1, 1024))
        inputs[f'y{i}'] = list(rng.uniform(-1, 1, 1024))
    enc = pub.encrypt(inputs, sig)
    enc.to_host(True)
    if world > 1:
        box = [{n: enc.get(n) for n in enc.names()} if dist.rank == 0 else None]
        tdd.broadcast_object_list(box, src=0)  # encryption draws fresh randomness: every rank works on rank 0's ciphertexts
        enc = SEALValuation()
        for n, (kind, size, limbs, scale, data) in box[0].items():
            enc._set_cipher(n, data, scale)
        attach_limb_dist(pub, dist)
    for _ in range(max(1, args.warmup)):
        out = pub.execute(compiled, enc)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = pub.execute(compiled, enc)
    dist.barrier()
    dt = dist.max_over_ranks(time.perf_counter() - t0)
    value = args.steps * n_products / dt
    ok = True
    if dist.rank == 0:
        from oracle import pyoracle as po  # checker only
        o = po.Oracle(N, list(pub.primes))
        ok = bool(np.array_equal(out.get('z0')[4], o.op_triple(enc.get('x0')[4], enc.get('y0')[4], pub.relin_key())))
    if dist.sum_over_ranks(0.0 if ok else 1.0) > 0:
        raise SystemExit("bench.py --shard limb: the sharded result differs from the CPU oracle — number withheld")
    if dist.rank == 0:
        xbytes = (l * N * 8) * (G - 1) / G + 2 * N * 8 + 2 * N * 8  # per GPU per triple: all-gather receive + two broadcasts
        kb = [int(b) for b in pub.key_bytes()]
        line = {"metric": "homomorphic ops/sec (mul+rescale+relin) at N=2^16, L=10; execute() wall-time",
                "value": round(value, 2), "unit": "op-triples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(dt * 1e3 / args.steps, 4), "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                "config": {"workload": f"op-triple multiply+relinearize+rescale, N=2^{args.logn}, L={l}, {n_products} triples per step "
                                       f"(one public_ctx.execute() of {n_products} products), each computed by all shards together",
                           "poly_modulus_degree": N, "limbs": l, "entry_point": "public_ctx.execute (shard_mode = 'limb')",
                           "parallelism": f"RNS limbs over {G} shard(s) on {world} GPU(s): limb i on shard i mod G; per key switch "
                                          "all-gather of the digits + broadcast of the special limb, per rescale one broadcast",
                           "ranks": world, "collectives_backend": dist.backend if world > 1 else None,
                           "exchange": (("RCCL (torch.distributed nccl) in place on the library's device buffers" if dist.backend == "nccl" else
                                         "gloo, staged through the host") if world > 1 else
                                        "one gather launch per receiving shard (peer reads between the shards' contexts, one GPU)"),
                           "exchange_bytes_per_gpu_per_triple": int(xbytes),
                           "exchange_launches_per_execute": int(pub.last_exchange_launches),
                           "exchanged_words_per_execute": int(pub.last_exchanged_words),
                           "key_bytes": {"whole_key": int(pub.relin_key().nbytes), "per_shard_here": kb[:-1]},
                           "includes": "input upload and output download (a limb-sharded value has no single device handle)"},
                "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "achieved": round(triple_bytes(N, l) * value / max(world, 1) / 1e9, 1),
                             "frac": round(triple_bytes(N, l) * value / max(world, 1) / 1e9 / HBM_PEAK_GBPS, 4),
                             "bytes_per_unit": triple_bytes(N, l),
                             "basis": "SURVEY.md 8(d) algorithmic bytes of one op-triple x op-triples/s, per GPU"},
                "verified": {"triples_checked": 1, "bit_exact_vs_oracle": True}, "cpu_baseline": None}
        print(json.dumps(line), flush=True)
    dist.close()


def subdag_leg(args, dist):
    """--shard subdag: ONE execute() of the Harris DAG (N = 2^15, L = 8) with its independent sub-DAGs on all
    GPUs of the job.  The mode lives inside public_ctx.execute (one process drives the
    devices, peer copies at the cuts — eva_amd/host/multi_device.h), so rank 0 computes on devices
    0..world-1 and the other ranks of a torchrun job only take part in the barriers."""
    import numpy as np
    import torch
    from eva.ckks import CKKSCompiler
    from eva.seal import generate_keys
    from eva_amd.roofline import dag_bytes, roofline as rl
    world = dist.world
    names = [torch.cuda.get_device_name(i) for i in range(torch.cuda.device_count())]
    line = None
    if dist.rank == 0:
        from eva_amd.workloads import harris as _harris, image as _image
        compiled, params, sig = CKKSCompiler(config={'warn_vec_size': 'false'}).compile(_harris())
        pad_chain(params, 9, 32768)
        ndev = torch.cuda.device_count()
        members = [r % ndev for r in range(world)] if world > 1 else [0] * max(2, args.shards)  # (gloo on a 1-GPU box: members share it)
        pub, sec = generate_keys(params, 1, devices=members, shard="subdag")
        nbytes, _ = dag_bytes(compiled, sig, 32768, 9)
        enc = pub.encrypt(_image(4096), sig)
        for _ in range(max(2, args.warmup)):
            out = pub.execute(compiled, enc)
        pub.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = pub.execute(compiled, enc)
        pub.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        from oracle.executor import c_walk  # checker only
        ref, _ = c_walk(pub, compiled, enc, threads=8)
        ok = all(np.array_equal(out.get(n)[4], ref[n]) for n in ref)
        if not ok:
            raise SystemExit("bench.py --shard subdag: the split execution differs from the CPU oracle — number withheld")
        line = {"metric": "execute() wall-time of one Harris DAG (N=2^15, L=8) split over the GPUs", "value": round(1.0 / dt, 2),
                "unit": "DAGs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 4),
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                "config": {"workload": "Harris corner detector, 64x64 image, N=2^15, L=8, independent sub-DAGs per device",
                           "members": members, "devices": names[:max(world, 1)],
                           "plan": [list(p) for p in pub.last_subdag_plan], "parallelism": f"sub-DAG split over {len(members)} member(s), peer copies at the cuts, one process"},
                "roofline": rl(nbytes, dt), "verified": {"bit_exact_vs_oracle": ok}, "cpu_baseline": None}
    dist.barrier()
    if line:
        print(json.dumps(line), flush=True)
    dist.close()


def _mem(tag, dev=0):
    """EVA_BENCH_MEMDEBUG=1: HBM in use at a point of the run (stderr)"""
    if os.environ.get("EVA_BENCH_MEMDEBUG") == "1":
        import torch
        free_b, total_b = torch.cuda.mem_get_info(dev)
        print(f"[mem] {tag}: {(total_b - free_b) / 1e9:.2f} GB in use", file=sys.stderr, flush=True)


def run_dag_legs(args):
    """the four DAG legs of the line (Harris latency, Harris batch, config 4, C1 / C2 / C5); a leg must not cost the line"""
    import gc
    legs = {}
    for name, fn in (("dag", lambda: dag_leg(15, host_cores())),
                     ("dag_harris_batch", lambda: dag_batch_leg(args.harris_batch, 5, workload="harris", chunk=args.harris_chunk or None)),
                     ("dag_batch", lambda: dag_batch_leg(args.dag_batch, 7)),
                     ("dag_configs", lambda: dag_configs_leg(9))):
        try:
            legs[name] = fn()
        except Exception as e:  # noqa: BLE001
            legs[name] = {"error": repr(e)}
        gc.collect()
    return legs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="independent op-triples per step")
    ap.add_argument("--logn", type=int, default=16)
    ap.add_argument("--limbs", type=int, default=10)
    ap.add_argument("--streams", type=int, default=2,
                    help="raw_cabi leg only: issue queues (forked contexts = HIP streams) its groups alternate between "
                         "(execute() alternates between the context's own two issue queues)")
    ap.add_argument("--group", type=int, default=32,
                    help="triples per execute() call = products of the compiled program (one wide launch set, shared key)")
    ap.add_argument("--separate-multiply", action="store_true",
                    help="raw_cabi leg only: evah_multiply_many + evah_relinearize_rescale_many instead of the one-call op-triple")
    ap.add_argument("--members", type=int, default=1,
                    help="--shard dag: contexts per rank sharing its GPU (shard_mode='dag' inside execute_batch)")
    ap.add_argument("--dag-batch", type=int, default=256, help="--shard dag / the dag_batch leg: independent Sobel DAGs in the batch")
    ap.add_argument("--harris-batch", type=int, default=64, help="the dag_harris_batch leg: independent Harris DAGs (N=2^15, L=8) in the batch")
    ap.add_argument("--harris-chunk", type=int, default=0, help="instances per batched device handle of the dag_harris_batch leg (0: the leg's default)")
    ap.add_argument("--legs-first", action="store_true", help="run the DAG legs before the headline (EVA_BENCH_LEGS_FIRST=1)")
    ap.add_argument("--only-leg", default="", help="run only this leg (dag | dag_batch | dag_harris_batch | dag_configs) and print its dict")
    ap.add_argument("--shard", choices=["ciphertexts", "limb", "subdag", "dag"], default="ciphertexts",
                    help="ciphertexts: independent triples per GPU, no collective (default, weak scaling); "
                         "limb: every triple on all GPUs, RNS limbs dealt over them (strong scaling); "
                         "subdag: one Harris execute() with its independent sub-DAGs on the GPUs")
    ap.add_argument("--shards", type=int, default=1, help="--shard limb with one process: shards on the one GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the raw C-ABI, host-valuation and DAG legs")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--raw-only", action="store_true",
                    help="run only the raw C-ABI leg (uploaded operands, no encrypt / decrypt kernels in the trace) and print its "
                         "dict: what scripts/collect_profiles.sh runs under the rocprofv3 counter passes — the launches per "
                         "group are the ones execute() issues")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    import numpy as np
    # torch first (inside Dist): the HIP runtime torch bundles must be the one every library shares
    from eva_amd.dist import Dist
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); "
                         "the product path has no CPU fallback")
    # nccl = RCCL, one GPU per rank (the driver's runs); gloo: ranks share the visible GPUs (1-GPU boxes, tests)
    dist = Dist(backend=os.environ.get("EVA_BENCH_BACKEND", "nccl"))
    rank, world, dev = dist.rank, dist.world, dist.device_index
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    if args.raw_only:
        from eva_amd.hostref import coeff_modulus_create
        N, l = 1 << args.logn, args.limbs
        print(json.dumps(raw_cabi_leg(args, N, l, coeff_modulus_create(N, [60] * (l + 1)), dev, args.steps, args.warmup)), flush=True)
        return dist.close()
    if args.only_leg:
        leg = {"dag": lambda: dag_leg(15, host_cores()), "dag_batch": lambda: dag_batch_leg(args.dag_batch, 7),
               "dag_harris_batch": lambda: dag_batch_leg(args.harris_batch, 5, workload="harris", chunk=args.harris_chunk or None),
               "dag_configs": lambda: dag_configs_leg(9)}[args.only_leg]()
        print(json.dumps({args.only_leg: leg}), flush=True)
        return dist.close()
    if args.shard == "limb":
        return limb_sharded(args, dist)
    if args.shard == "subdag":
        return subdag_leg(args, dist)
    if args.shard == "dag":
        return dag_sharded(args, dist)
    # what every rank runs on: the driver's multi-GPU runs show that RCCL saw N ranks on N devices
    dev_name = torch.cuda.get_device_name(dev)
    rank_devices = [dev_name]
    if world > 1:
        import torch.distributed as tdist
        gathered = [None] * world
        tdist.all_gather_object(gathered, f"rank {rank}: cuda:{dev} {dev_name}")
        rank_devices = gathered

    from eva.seal import generate_keys

    # --legs-first / EVA_BENCH_LEGS_FIRST=1: the DAG legs before the headline's state exists (each leg builds and frees its
    # own key pair and valuations; the order of the legs inside one process is not part of any number's definition)
    dag_legs_first = None
    if (args.legs_first or os.environ.get("EVA_BENCH_LEGS_FIRST") == "1") and world == 1 and not args.no_legs:
        hold = None
        if os.environ.get("EVA_BENCH_HOLD_GB"):  # experiment: other allocations of the process alive during the legs
            hold = [torch.empty(1 << 28, dtype=torch.uint8, device=f"cuda:{dev}") for _ in range(4 * int(os.environ["EVA_BENCH_HOLD_GB"]))]
        dag_legs_first = run_dag_legs(args)
        hold = None

    N, l = 1 << args.logn, args.limbs
    k = l + 1
    G = max(1, min(args.group, 64, args.batch))
    n_calls = max(1, args.batch // G)          # execute() calls per step
    batch = n_calls * G                        # triples per step
    compiled, params, sig = triple_program(G, N, l)
    _mem("before the headline's keys", dev)
    pub, sec = generate_keys(params, 17 + rank)
    pub.device = dev
    sec.device = dev
    primes = list(pub.primes)

    # synthetic inputs: every triple of a step has its own operand pair (distinct HBM data: no triple finds its
    # inputs in cache because another used them), encrypted by this key pair and LEFT IN HBM by encrypt()
    rng = np.random.default_rng(0xE7A + rank)
    clear, vals = [], []
    for _ in range(n_calls):
        inputs = {}
        for i in range(G):
            inputs[f'x{i}'] = list(rng.uniform(-1, 1, 1024))
            inputs[f'y{i}'] = list(rng.uniform(-1, 1, 1024))
        clear.append(inputs)
        vals.append(pub.encrypt(inputs, sig))
    st0 = pub.transfer_stats()

    def step():
        return [pub.execute(compiled, v) for v in vals]

    def barrier():
        pub.synchronize()
        dist.barrier()  # torch.cuda.synchronize() + barrier over the ranks

    outs = None
    for _ in range(max(args.warmup, 2)):       # device context / key upload / pools of both issue queues
        outs = step()
    barrier()
    pub.profile(True)                          # HIP-event brackets around every launch of the timed region
    pub.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        outs = step()
    barrier()
    dt = time.perf_counter() - t0
    dt = dist.max_over_ranks(dt)
    prof = {c: v for c, v in pub.profile_get().items() if v[0]}
    st1 = pub.transfer_stats()
    moved = {kk: int(st1[kk] - st0[kk]) for kk in ("ct_uploads", "ct_downloads")}

    triples = args.steps * batch * world
    value = triples / dt

    # The calls alternate between two issue queues whose launches overlap, so a launch's HIP-event duration in the
    # timed region is the time it SHARED the GPU for.  Outside the timed region: calls separated by a synchronize,
    # every launch bracketed, for the duration a kernel takes with the GPU to itself (what a rocprofv3 PMC pass,
    # which serialises the kernels, reports as well).
    pub.profile_reset()
    for _ in range(2):
        for v in vals:
            outs_alone = pub.execute(compiled, v)
            pub.synchronize()
    prof_excl = {c: v for c, v in pub.profile_get().items() if v[0]}
    pub.profile(False)
    del outs_alone

    # ---- outside the timed region: the first and last product of every group of the LAST timed step are downloaded and
    # compared with the CPU oracle's op-triple of the same encrypted operands and key; the same outputs are decrypted
    verified, host_ops = None, []
    from oracle import pyoracle as po  # checker only
    o = po.Oracle(N, primes)
    ok, checked, max_err = True, 0, 0.0
    key_host = pub.relin_key()
    for ci, (v, out) in enumerate(zip(vals, outs)):
        dec = sec.decrypt(out, sig)
        for i in sorted({0, G - 1}):
            a, b = v.get(f'x{i}')[4], v.get(f'y{i}')[4]
            if len(host_ops) < 4:
                host_ops.append((a, b))
            ok = ok and bool(np.array_equal(out.get(f'z{i}')[4], o.op_triple(a, b, key_host)))
            want = np.asarray(clear[ci][f'x{i}']) * np.asarray(clear[ci][f'y{i}'])
            max_err = max(max_err, float(np.max(np.abs(np.asarray(dec[f'z{i}']) - want))))
            checked += 1
    bad = dist.sum_over_ranks(0.0 if ok and max_err < 1e-3 else 1.0)
    if bad > 0:
        raise SystemExit("bench.py: the timed path's output differs from the CPU oracle (or does not decrypt to x * y) — number withheld")
    verified = {"triples_checked": checked, "bit_exact_vs_oracle": True, "decrypt_max_abs_err_vs_x_times_y": max_err,
                "what": "outputs of the last timed step's execute() calls vs oracle multiply+relinearize+rescale of the same ciphertexts and key"}

    # N > 1: BASELINE config 4's scaling leg rides along — every rank runs its share (instance b on rank b mod
    # world) of the 256 Sobel DAGs, so a multi-GPU run of the default command reports DAGs/s and the achieved-HBM
    # fraction per GPU count as well (all ranks take part: the timing is barrier-bracketed)
    dag_batch_multi = None
    if world > 1 and not args.no_legs:
        dag_batch_multi = dag_batch_leg(args.dag_batch, 5, dist)

    if rank == 0:
        cb = class_bytes(N, l, k, G)
        dom = max(prof, key=lambda c: prof[c][1]) if prof else None
        tb = triple_bytes(N, l)
        key_bytes_once = 2 * l * (l + 1) * N * 8
        roofline = {"bound": "valu", "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "achieved": round(tb * value / world / 1e9, 1),
                    "frac": round(tb * value / world / 1e9 / HBM_PEAK_GBPS, 4),
                    "hbm_frac": round(tb * value / world / 1e9 / HBM_PEAK_GBPS, 4),
                    "bytes_per_unit": tb,
                    "basis": "achieved / peak / frac: SURVEY.md 8(d) algorithmic bytes of one op-triple (inputs, outputs and key "
                             "read/written once; NTT-internal passes count as zero) x op-triples/s per GPU over the HBM peak.  "
                             "bound = valu: the SQ counters put the VALUs at valu_frac of the time against hbm_frac of the "
                             "HBM peak (64-bit modular multiplies, DESIGN.md section 4)",
                    # 8(d) charges the key to every triple; a launch set of G triples reads it once
                    "launch_compulsory_bytes": int(G * (tb - key_bytes_once) + key_bytes_once),
                    "launch_compulsory_frac": round((G * (tb - key_bytes_once) + key_bytes_once) * (value / world / G) / 1e9 / HBM_PEAK_GBPS, 4),
                    "launch_set": f"{G} triples per execute(): one relinearization key read per launch set"}
        if dom:
            n_l, ms = prof[dom]
            avg_us = ms * 1e3 / max(n_l, 1)
            kern_total_ms = sum(v[1] for v in prof.values())
            traffic = None  # PMC bytes per launch of the dominant kernel, from the committed rocprofv3 passes
            tpath = os.path.join(ROOT, "profiles", "bench_pmc_traffic.json")
            if os.path.exists(tpath):
                try:
                    from eva_amd.roofline import csrc_tree_hash
                    tj = json.load(open(tpath))
                    traffic = tj["by_class"].get(dom, {}).get("hbm_bytes_per_launch")
                    # the counter passes are a separate, committed measurement: tie them to the tree being timed
                    roofline["traffic_source"] = {"file": "profiles/bench_pmc_traffic.json", "tree": tj.get("tree"),
                                                  "commit": tj.get("commit"), "tree_now": csrc_tree_hash(ROOT)}
                    roofline["traffic_stale"] = tj.get("tree") != csrc_tree_hash(ROOT)
                except Exception:
                    traffic = None
            roofline["kernel"] = dom
            roofline["traffic"] = traffic
            roofline["dominant"] = {
                "kernel": dom, "avg_launch_us": round(avg_us, 2), "launches_sampled": n_l,
                "distinct_bytes_per_launch": int(cb.get(dom, 0)),
                "achieved": round(cb.get(dom, 0) / (avg_us * 1e-6) / 1e9, 1) if avg_us else None,
                "sampling": f"HIP events on the launch stream around every launch inside the timed region ({G} triples per launch)"}
            if prof_excl.get(dom, (0, 0))[0]:
                ex_us = prof_excl[dom][1] * 1e3 / prof_excl[dom][0]
                roofline["dominant"].update({
                    "concurrent_queues": 2,
                    "avg_launch_us_alone": round(ex_us, 2), "launches_sampled_alone": prof_excl[dom][0],
                    "achieved_alone": round(cb.get(dom, 0) / (ex_us * 1e-6) / 1e9, 1),
                    "note": ("execute() alternates between 2 issue queues: launches of consecutive calls overlap in the timed region, so "
                             "avg_launch_us is the time a launch shared the GPU for; *_alone = the same launches with a synchronize "
                             "between the calls, measured right after the timed region")})
                roofline["by_class_us_alone"] = {c: round(v[1] * 1e3 / max(v[0], 1), 2) for c, v in prof_excl.items() if v[0]}
            roofline["by_class_us"] = {c: round(v[1] * 1e3 / max(v[0], 1), 2) for c, v in prof.items() if v[0]}
            roofline["by_class_share"] = {c: round(v[1] / kern_total_ms, 3) for c, v in prof.items() if v[0]}

        # the limiter (SURVEY 8(d) "secondary"): 32-bit integer-multiply / VALU issue.  The instruction count per
        # op-triple comes from the committed SQ-counter pass of this same command
        # (profiles/bench_valu_issue.json); the share of the measured time the VALUs spent issuing is
        # that count's issue time over this run's time per triple.
        vpath = os.path.join(ROOT, "profiles", "bench_valu_issue.json")
        if os.path.exists(vpath) and (args.logn, l) == (16, 10):
            try:
                vj = json.load(open(vpath))
                from eva_amd.roofline import csrc_tree_hash
                us_per_triple = 1e6 * world / value
                roofline["valu_frac"] = round(vj["valu_issuing_us_per_triple"] / us_per_triple, 3)
                roofline["secondary"] = {
                    "bound": "valu integer issue", "valu_wave_instructions_per_triple": vj["valu_wave_instructions_per_triple"],
                    "valu_issuing_us_per_triple": vj["valu_issuing_us_per_triple"], "us_per_triple": round(us_per_triple, 2),
                    "frac": round(vj["valu_issuing_us_per_triple"] / us_per_triple, 3),
                    "stale": vj.get("tree") != csrc_tree_hash(ROOT),
                    "note": "9 integer multiplies + 8 other VALU instructions per butterfly at ~4.5-5.6 SIMD-cycles each: "
                            "the path is issue-bound before it is HBM-bound (DESIGN.md section 4)"}
            except Exception:
                pass

        legs = {}
        if world == 1 and not args.no_legs:
            try:
                legs["raw_cabi"] = raw_cabi_leg(args, N, l, primes, dev, max(5, min(args.steps, 20)), 3)
            except Exception as e:  # noqa: BLE001 — a leg must not cost the headline line
                legs["raw_cabi"] = {"error": repr(e)}
            try:
                want = outs[0].get('z0')[4]
                legs["execute_path"] = dict({"program": f"{G} independent products z_i = x_i * y_i (Mul -> Relinearize -> Rescale), "
                                                        f"N=2^{args.logn}, L={l}: the headline's program with host valuations"},
                                            **host_valuation_legs(pub, compiled, vals[0], G, l, N, 12, want))
            except Exception as e:  # noqa: BLE001
                legs["execute_path"] = {"error": repr(e)}
            # the headline's key pair, valuations (2.7 GB of operands) and outputs are not needed past this point
            _mem("after the headline and its side legs", dev)
            # (the verification loop's variables hold valuations too — device handles that keep the key pair's pools alive:
            # 5.9 GB stayed allocated through the DAG legs until r6, and config 4 ran 9 % slower beside them)
            outs = vals = pub = sec = v = out = dec = None
            import gc
            gc.collect()
            _mem("after dropping the headline's state", dev)
            try:
                free_b, total_b = torch.cuda.mem_get_info(dev)
                legs["hbm_in_use_before_dag_legs_gb"] = round((total_b - free_b) / 1e9, 2)
            except Exception:  # noqa: BLE001
                pass
            legs.update(dag_legs_first if dag_legs_first is not None else run_dag_legs(args))
        if dag_batch_multi is not None:
            legs["dag_batch"] = dag_batch_multi

        cpu = None
        if not args.no_cpu_baseline and world == 1:  # reported baseline: rank 0 at N=1 only
            hp = host_ops
            a, b = hp[0]
            t1 = time.perf_counter()
            o.op_triple(a, b, key_host)
            one = time.perf_counter() - t1
            n = max(1, min(50, int(args.cpu_seconds / max(one, 1e-3))))
            t1 = time.perf_counter()
            for i in range(n):
                a, b = hp[i % len(hp)]
                o.op_triple(a, b, key_host)
            cdt = time.perf_counter() - t1
            # the same port on many host cores at once (independent triples, one per thread; ctypes
            # releases the GIL) — the analogue of the reference's Galois node-level parallelism
            import threading

            def concurrent(threads):
                """`threads` op-triples, one per thread, at once -> op-triples/s"""
                done = []

                def worker(i):
                    a_, b_ = hp[i % len(hp)]
                    o.op_triple(a_, b_, key_host)
                    done.append(i)
                ths = [threading.Thread(target=worker, args=(i,)) for i in range(threads)]
                t1_ = time.perf_counter()
                for t in ths:
                    t.start()
                for t in ths:
                    t.join()
                return len(done) / (time.perf_counter() - t1_)
            # every core the process may run on (the reference sizes its pool from the affinity mask,
            # /root/reference/python/eva/__init__.py:10-14), and 64 threads beside it when the host has more
            cores = host_cores()
            all_rate = concurrent(cores)
            rate64 = concurrent(64) if cores > 64 else None
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            try:
                import seal_probe
                seal = seal_probe.probe()
            except Exception:  # noqa: BLE001
                seal = {"present": False}
            cpu = {"value": round(n / cdt, 3), "unit": "op-triples/s", "cores": 1, "kind": "port",
                   "seal": "present: " + ",".join(seal.get("paths", [])[:2]) if seal.get("present") else
                           "SEAL absent on this host (tools/seal_probe.py): the oracle restatement is the baseline",
                   "pin_with_seal": "on a host with Microsoft SEAL >= 3.6: bash tools/pin_with_seal.sh   (exports the vectors, builds "
                                    "tools/seal_parity.cpp against find_package(SEAL), diffs primes, psi, NTT, every evaluator call, the "
                                    "op-triple, encode, decrypt, decode and the object format with SEAL's own bits, times SEAL's op-triple; "
                                    "writes profiles/seal_pin.json, which this file then reports as cpu_baseline.kind = 'reference')",
                   "all_cores": {"value": round(all_rate, 2), "cores": cores,
                                 "sample": f"{cores} op-triples, one per thread, concurrently, on all {cores} cores of the "
                                           "affinity mask"},
                   "threads_64": None if rate64 is None else {"value": round(rate64, 2), "cores": 64,
                                                               "sample": "64 op-triples, one per thread, concurrently"},
                   "sample": f"{n} op-triples (multiply+relinearize+rescale) at N=2^{args.logn}, "
                             f"L={l}, same ciphertexts/key as the GPU run, oracle/libeva_oracle.so, "
                             f"1 thread of {cores} host cores"}
            cpu = seal_pin_baseline(cpu, N, l)
        line = {
            "metric": "homomorphic ops/sec (mul+rescale+relin) at N=2^16, L=10; execute() wall-time",
            "value": round(value, 2), "unit": "op-triples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt * 1e3 / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"op-triple multiply+relinearize+rescale, N=2^{args.logn}, "
                                   f"L={l} data limbs + 1 special prime (60-bit), "
                                   f"{batch} independent triples per step per GPU",
                       "poly_modulus_degree": N, "limbs": l, "batch_per_gpu": batch,
                       "entry_point": "public_ctx.execute",
                       "execute_calls_per_step": n_calls, "triples_per_call": G,
                       "program": f"{G} products z_i = x_i * y_i, compiled with eager relinearization: Mul -> Relinearize -> Rescale each",
                       "valuations": "device-resident (encrypt -> execute -> decrypt by handle): inputs are in HBM when the timed "
                                     "region starts, execute() enqueues and returns, the timed region ends with synchronize()",
                       "ciphertexts_moved_over_pcie_in_timed_region": moved,
                       "issue_queues_per_gpu": 2,
                       "parallelism": f"independent ciphertexts sharded over {world} GPU(s), no collective",
                       "ranks": world, "collectives_backend": dist.backend,
                       "rccl_ranks": world if dist.backend == "nccl" else 0, "rank_devices": rank_devices,
                       "collectives": "barrier + max-over-ranks of the wall time (torch.distributed "
                                      + ("nccl = RCCL" if dist.backend == "nccl" else dist.backend) + "); none in the data path"},
            "roofline": roofline, "verified": verified, "cpu_baseline": cpu,
        }
        line.update(legs)
        print(json.dumps(line), flush=True)
    dist.close()


def seal_pin_baseline(cpu, N, l):
    """tools/pin_with_seal.sh leaves profiles/seal_pin.json on a host that has Microsoft SEAL >= 3.6: the verdicts of
    tools/seal_parity.cpp's sections and SEAL's own op-triple rate there.  When it is present (and was measured at this
    (N, L)) the reported baseline is the reference itself — kind "reference" — with the port's figures kept beside it."""
    path = os.environ.get("EVA_SEAL_PIN_JSON") or os.path.join(ROOT, "profiles", "seal_pin.json")
    if not os.path.exists(path):
        return cpu
    try:
        pin = json.load(open(path))
        rate = pin.get("seal_triples_per_s")
        if pin.get("dry_run") or not rate or (pin.get("N"), pin.get("limbs")) != (N, l):
            cpu["seal_pin"] = {"file": "profiles/seal_pin.json", "used": False,
                               "why": "a dry run: no SEAL took part" if pin.get("dry_run") else "no op-triple rate at this (N, L) in it"}
            return cpu
        port = {kk: cpu[kk] for kk in ("value", "cores", "sample", "all_cores", "threads_64")}
        cpu.update({"value": round(float(rate), 3), "cores": int(pin.get("cores", 1)), "kind": "reference",
                    "sample": pin.get("sample", "Evaluator::multiply + relinearize_inplace + rescale_to_next_inplace, tools/seal_parity.cpp --time-triple"),
                    "seal": f"Microsoft SEAL {pin.get('seal_version', '?')} (tools/pin_with_seal.sh on {pin.get('host', '?')})",
                    "port": port, "seal_pin": {"file": "profiles/seal_pin.json", "used": True, "sections": pin.get("sections"),
                                               "all_sections_identical": pin.get("all_identical")}})
    except Exception as e:  # noqa: BLE001
        cpu["seal_pin"] = {"file": "profiles/seal_pin.json", "used": False, "why": repr(e)}
    return cpu


if __name__ == "__main__":
    main()
