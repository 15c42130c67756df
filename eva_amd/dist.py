"""One-process-per-GPU sharding helpers (torch.distributed; backend "nccl" = RCCL on ROCm,
"gloo" on CPU).

The execute() path shards over independent units — ciphertext op batches or whole DAG instances
(BASELINE config 4; SURVEY.md §8e "instance b -> GPU b mod G") — so there is no data-path
collective: every rank owns its keys and inputs, the only communication is the barrier, the
max-over-ranks wall time and, optionally, gathering per-unit results on rank 0.
"""
import os
import time


class Dist:
    def __init__(self, backend=None, timeout_s=600):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
        self.on_gpu = torch.cuda.is_available()
        if backend is None:
            backend = "nccl" if self.on_gpu else "gloo"
        self.backend = backend
        self.owns_group = False
        # one GPU per rank under RCCL; under gloo several ranks may share a device (tests on a 1-GPU box)
        self.device_index = self.local_rank
        if self.on_gpu and backend != "nccl":
            self.device_index = self.local_rank % max(torch.cuda.device_count(), 1)
        if self.on_gpu:
            torch.cuda.set_device(self.device_index)
        if self.world > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            import datetime
            kw = {}
            if backend == "nccl":
                kw["device_id"] = torch.device("cuda", self.local_rank)
            dist.init_process_group(backend, rank=self.rank, world_size=self.world,
                                    timeout=datetime.timedelta(seconds=timeout_s), **kw)
            self.owns_group = True

    # ---- partitioning
    def my_units(self, n_units):
        """Unit b runs on rank b mod world (keeps every rank's share within one unit of equal)."""
        return list(range(self.rank, n_units, self.world))

    # ---- collectives used around (never inside) the data path
    def _device(self):
        return "cuda" if (self.on_gpu and self.backend == "nccl") else "cpu"

    def barrier(self):
        if self.on_gpu:
            self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()

    def max_over_ranks(self, x):
        if self.world == 1:
            return float(x)
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self._device())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x):
        if self.world == 1:
            return float(x)
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self._device())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def gather_units(self, local, n_units):
        """local: {unit index: picklable result} of this rank -> on rank 0 the list of all
        n_units results in unit order (None elsewhere)."""
        if self.world == 1:
            return [local[i] for i in range(n_units)]
        parts = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object(local, parts, dst=0)
        if self.rank != 0:
            return None
        merged = {}
        for p in parts:
            merged.update(p)
        return [merged[i] for i in range(n_units)]

    def timed(self, fn):
        """barrier; fn(); barrier -> max over ranks of the wall time."""
        self.barrier()
        t0 = time.perf_counter()
        out = fn()
        self.barrier()
        return out, self.max_over_ranks(time.perf_counter() - t0)

    def close(self):
        if self.owns_group and self.dist.is_initialized():
            self.dist.destroy_process_group()


def run_sharded(dist, n_units, work):
    """Run work(unit) for this rank's units; returns (results on rank 0 | None, max wall seconds)."""
    def body():
        return {u: work(u) for u in dist.my_units(n_units)}
    local, secs = dist.timed(body)
    return dist.gather_units(local, n_units), secs


class _DevWords:
    """a library device buffer (pointer, words) as a CUDA array for torch.as_tensor"""

    def __init__(self, ptr, words):
        self.__cuda_array_interface__ = {"shape": (int(words),), "typestr": "<i8", "data": (int(ptr), False), "version": 2}


def attach_limb_dist(pub, dist):
    """RNS-limb sharding of public_ctx.execute() across the ranks of `dist` (one process per GPU; SURVEY.md 8(e) row 3,
    north_star: "RCCL all-gather over xGMI reassembling limbs before each key-switch").  The C++ limb-shard evaluator
    behind execute() (eva_amd/host/multi_device.h) keeps this rank's limbs and calls back here at its exchange steps:
      nccl (= RCCL)  collectives in place on the library's device buffers.  A torch stream created here becomes BOTH the
                     stream the shard's kernels run on (set_limb_dist hands its handle to the library) and torch's
                     current stream around every collective, so producer kernel -> collective -> consumer kernel are
                     ordered by the streams alone and nothing waits for the host between phases.  (torch's default
                     stream would not do: its handle is 0, which the library reads as "keep your own stream", and a
                     library stream is non-blocking — unordered against the null stream the collectives would use.)
      gloo           the same steps staged through host memory (tests: several ranks sharing one GPU)"""
    import numpy as np
    torch, tdist = dist.torch, dist.dist
    device_collectives = dist.backend == "nccl"
    rank, world = dist.rank, dist.world
    stream = torch.cuda.Stream(device=dist.device_index) if device_collectives else None
    if stream is not None and not stream.cuda_stream:
        raise RuntimeError("attach_limb_dist: torch returned the null stream; the limb shard needs a stream of its own")

    def view(ptr, words):
        return torch.as_tensor(_DevWords(ptr, words), device=f"cuda:{dist.device_index}")

    def all_gather(ptr, chunk):
        if device_collectives:
            with torch.cuda.stream(stream):
                full = view(ptr, world * chunk)
                tdist.all_gather_into_tensor(full, full[rank * chunk:(rank + 1) * chunk])
            return
        full = view(ptr, world * chunk)
        torch.cuda.synchronize()  # the library's own stream has produced the chunk
        mine = full[rank * chunk:(rank + 1) * chunk].cpu()
        parts = [torch.empty_like(mine) for _ in range(world)]
        tdist.all_gather(parts, mine)
        for r, p in enumerate(parts):
            if r != rank:
                full[r * chunk:(r + 1) * chunk].copy_(p)
        torch.cuda.synchronize()

    def broadcast(ptr, words, owner):
        if device_collectives:
            with torch.cuda.stream(stream):
                tdist.broadcast(view(ptr, words), src=owner)
            return
        t = view(ptr, words)
        torch.cuda.synchronize()
        h = t.cpu() if rank == owner else torch.empty(words, dtype=torch.int64)
        tdist.broadcast(h, src=owner)
        if rank != owner:
            t.copy_(h)
        torch.cuda.synchronize()

    def sum_host(arr):
        t = torch.from_numpy(np.asarray(arr).view(np.int64))  # shares memory: the all-reduce lands in the caller's words
        if device_collectives:
            with torch.cuda.stream(stream):
                d = t.to(f"cuda:{dist.device_index}")
                tdist.all_reduce(d)
                t.copy_(d.cpu())  # .cpu() waits for the stream
        else:
            tdist.all_reduce(t)

    dist._limb_streams = getattr(dist, "_limb_streams", []) + [stream]  # the library holds the raw handle: keep it alive
    pub.set_limb_dist(rank, world, all_gather, broadcast, sum_host, stream.cuda_stream if stream is not None else 0)
