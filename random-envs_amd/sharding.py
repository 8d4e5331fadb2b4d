"""Multi-GPU plumbing: one process per GPU, the env batch sharded by index, no data-path collective.

The reference has nothing distributed; envs are independent objects, so a
batch shards trivially: rank r owns the contiguous global indices [r*B, (r+1)*B).  RNG streams are
keyed by GLOBAL env index inside the kernels, so results do not depend on the sharding.  The only
collectives are a SUM of the step counter and a MAX of the elapsed time (RCCL over xGMI via
torch.distributed backend "nccl"; "gloo" on CPU for tests)."""
import os


def dist_env():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard(batch_per_rank, rank):
    """(env_offset, batch) of this rank under weak scaling."""
    return rank * batch_per_rank, batch_per_rank


def shard_strong(global_batch, rank, world):
    """(env_offset, batch) of this rank when a fixed global batch is split (remainder to low ranks)."""
    q, r = divmod(global_batch, world)
    b = q + (1 if rank < r else 0)
    off = rank * q + min(rank, r)
    return off, b


def init(backend, device=None):
    import torch.distributed as dist
    rank, local_rank, world = dist_env()
    # REX_FORCE_DIST=1: initialise the process group at world size 1 too (exercises the RCCL path on a one-GPU box)
    if (world > 1 or os.environ.get("REX_FORCE_DIST")) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, **kw)
    return rank, local_rank, world


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def reduce_counter_and_time(steps_done, elapsed_s, device):
    """SUM of env-steps over ranks and MAX of the elapsed time (the only collectives of the path)."""
    import torch
    import torch.distributed as dist
    c = torch.tensor([int(steps_done)], dtype=torch.int64, device=device)
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(c.item()), float(t.item())


class StepCounter:
    """The path's ONLY collective: the global env-step counter, summed over ranks
    asynchronously every `every` steps so it never sits on the step critical path.  `add()` is called once per
    batched step; every `every`-th call snapshots the local count into a staging tensor and issues
    ``all_reduce(SUM, async_op=True)`` (RCCL runs it on its own stream; gloo on its worker thread); the previous
    reduction is waited for only when the next one is issued.  `total()` drains and returns the exact global
    count.  At world size 1 it is a plain integer."""

    def __init__(self, device, every=256):
        import torch
        import torch.distributed as dist
        self._dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.every = max(int(every), 1)
        self.local = 0            # env-steps of this rank
        self.n_calls = 0
        self.last_global = 0      # most recent completed global sum (lags by < 2 * every steps)
        self._work = None
        self._buf = torch.zeros(1, dtype=torch.int64, device=device)
        self.reductions = 0

    def add(self, env_steps):
        self.local += int(env_steps)
        self.n_calls += 1
        if self._dist is not None and self.n_calls % self.every == 0:
            self._issue()

    def _issue(self):
        self._drain()
        self._buf.fill_(self.local)
        self._work = self._dist.all_reduce(self._buf, op=self._dist.ReduceOp.SUM, async_op=True)
        self.reductions += 1

    def _drain(self):
        if self._work is not None:
            self._work.wait()
            self.last_global = int(self._buf.item())
            self._work = None

    def total(self):
        if self._dist is None:
            return self.local
        self._issue()
        self._drain()
        return self.last_global


def reduce_max(value, device):
    """MAX over ranks of a host float (the elapsed time of the bench contract)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shutdown():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
