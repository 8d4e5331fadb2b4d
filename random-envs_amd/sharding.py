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


def world_size():
    """Ranks of the initialised process group (what RCCL / gloo actually sees), else WORLD_SIZE from the environment."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size()
    return dist_env()[2]


def shard(batch_per_rank, rank):
    """(env_offset, batch) of this rank under weak scaling."""
    return rank * batch_per_rank, batch_per_rank


WAVE_ENVS = 64   # envs per wavefront of the step kernels at their widest (one lane per env, past 32 768 envs per GPU; 32 with two lanes per env):
                 # the solver instantiation is picked per wave


def shape_for_batch(kind, batch, simds=1024):
    """The launch shape rex_create picks for `batch` envs of `kind` on a GPU with `simds` SIMDs (MI355X: 1 024): the rule of
    rex_hip.hip::create_body restated for the host (tests/test_gpu_api.py::test_launch_shape_follows_the_batch holds the two together)."""
    planar = kind in ("hopper", "halfcheetah", "walker2d")
    pair = bool(planar and batch <= 32 * simds)
    lanes = 64 if batch > 32 * simds else 32
    if pair:   # two lanes per env: 64-lane blocks; walker2d / half-cheetah batches of >= 8 envs per SIMD halve them while the halved
        lanes = 64   # blocks still number <= SIMDs, down to 16 lanes (pair_lanes_for)
        while kind != "hopper" and batch >= 8 * simds and lanes > 16 and (4 * batch + lanes - 1) // lanes <= simds:
            lanes //= 2
    return dict(lanes=lanes, pair=pair,
                rolled=bool(kind == "hopper" and batch > 64 * simds), hum_pair=(kind == "humanoid"))


def pin_global_shape(env, global_batch, simds=None):
    """Give a shard the launch shape its GLOBAL batch would get on one GPU (rex_set_launch_shape).  rex_create picks the shape from
    the per-GPU batch: 65 536 envs on one GPU run one lane per env, the same envs split over two GPUs two lanes per env, and the two
    kernels round differently.  Pinned, an index-sharded run reproduces the single-GPU trajectories bit for bit (given shard
    boundaries on whole waves: shard_strong); unpinned, every shard runs the fastest shape for its own size and agrees with the
    single-GPU run to fp32 rounding only."""
    if simds is None:
        import torch
        simds = 4 * torch.cuda.get_device_properties(env.device).multi_processor_count
    sh = shape_for_batch(env.kind, int(global_batch), simds)
    kw = dict(lanes=sh["lanes"])
    if env.kind in ("hopper", "halfcheetah", "walker2d"):
        kw.update(pair=sh["pair"])
    if env.kind == "hopper":
        kw.update(rolled=sh["rolled"])
    return env.set_launch_shape(**kw)


def shard_strong(global_batch, rank, world, align=WAVE_ENVS):
    """(env_offset, batch) of this rank when a fixed global batch is split.  Shard boundaries fall on multiples of `align`
    envs (one wavefront of the step kernels at its widest) wherever the batch allows it, so that every wave holds the same envs
    as in the single-GPU run.  Everything the RNG decides is bit-identical under any split.  The ARITHMETIC of a lane also depends
    on which solver instantiation its wave picks and on the launch shape rex_create derives from the PER-GPU batch, so an
    index-sharded run reproduces the single-GPU trajectories bit for bit only when the shards also run the global batch's shape
    (`pin_global_shape`, `bench.py --pin-shape`) or under REX_FAST=0; otherwise lanes agree to fp32 rounding.  The remainder goes to
    the low ranks, the last rank takes what is left."""
    align = max(int(align), 1)
    blocks, tail = divmod(global_batch, align)
    if blocks < world:           # fewer whole waves than ranks: plain split
        q, r = divmod(global_batch, world)
        return rank * q + min(rank, r), q + (1 if rank < r else 0)
    q, r = divmod(blocks, world)
    nb = q + (1 if rank < r else 0)
    off = (rank * q + min(rank, r)) * align
    b = nb * align + (tail if rank == world - 1 else 0)
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
    asynchronously every `every` steps and never on the step's critical path.  `add()` is called once per batched
    step; every `every`-th call writes the HOST-side local count into one of two staging tensors and issues
    ``all_reduce(SUM, async_op=True)`` on a side stream of its own (RCCL orders the collective behind that stream, not
    behind the compute stream; gloo runs it on its worker thread).  Nothing is read back while stepping: the host waits
    only for the reduction issued TWO rounds earlier (the one whose staging tensor it is about to reuse -- long
    finished), and the compute stream is never made to wait.  `total()` issues a last reduction, drains and returns
    the exact global count.  At world size 1 it is a plain integer."""

    def __init__(self, device, every=256):
        import torch
        import torch.distributed as dist
        self._torch = torch
        self._dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.every = max(int(every), 1)
        self.local = 0            # env-steps of this rank
        self.n_calls = 0
        self.last_global = 0      # result of the most recent reduction that total() / poll() has read back
        self._works = [None, None]
        self._side = torch.cuda.Stream(device=device) if torch.device(device).type == "cuda" else None
        with self._ctx():         # staging buffers are created AND zero-filled on the side stream: every later fill_ / all_reduce on
            self._bufs = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(2)]   # them is ordered behind that fill
        self._polled = [True, True]   # staging buffer already folded into last_global
        self.reductions = 0

    def add(self, env_steps):
        self.local += int(env_steps)
        self.n_calls += 1
        if self._dist is not None and self.n_calls % self.every == 0:
            self._issue()

    def _ctx(self):
        import contextlib
        return self._torch.cuda.stream(self._side) if self._side is not None else contextlib.nullcontext()

    def _issue(self):
        k = self.reductions % 2
        with self._ctx():
            if self._works[k] is not None:     # two rounds old: complete long ago (RCCL: a stream-side wait on the side stream)
                self._works[k].wait()
            self._bufs[k].fill_(self.local)
            self._works[k] = self._dist.all_reduce(self._bufs[k], op=self._dist.ReduceOp.SUM, async_op=True)
        self._polled[k] = False
        self.reductions += 1
        return k

    def poll(self):
        """Non-blocking: the lagging global count.  Folds into `last_global` every reduction that has COMPLETED since the last call
        (`work.is_completed()`; nothing is waited for, the compute stream is not touched) and returns it; 0 until the first one lands."""
        if self._dist is None:
            self.last_global = self.local
            return self.last_global
        for k in (0, 1):
            w = self._works[k]
            if w is not None and not self._polled[k] and w.is_completed():
                with self._ctx():
                    self.last_global = max(self.last_global, int(self._bufs[k].item()))
                self._polled[k] = True
        return self.last_global

    def total(self):
        if self._dist is None:
            return self.local
        k = self._issue()
        with self._ctx():
            self._works[k].wait()
            self.last_global = int(self._bufs[k].item())   # the only blocking read-back: after the timed region
        self._polled[k] = True
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
