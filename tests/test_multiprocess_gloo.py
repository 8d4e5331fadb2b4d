"""N > 1 launch path on CPU: world_size 2, gloo.  Covers the sharding arithmetic and the only
collectives the path uses (SUM of the step counter -- issued asynchronously every K steps -- and MAX of the elapsed time)."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, json
sys.path.insert(0, %r)
import torch
from random_envs_amd import sharding
rank, local_rank, world = sharding.init("gloo")
assert world == 2
off, b = sharding.shard(32768, rank)
assert (off, b) == (rank * 32768, 32768)
so, sb = sharding.shard_strong(32769, rank, world)
steps = 10 * b + rank                      # each rank "did" a different amount of work
total, tmax = sharding.reduce_counter_and_time(steps, 1.0 + rank, torch.device("cpu"))
# the asynchronous counter of bench.py: 25 batched steps, an all-reduce in flight every 8 of them
ctr = sharding.StepCounter(torch.device("cpu"), every=8)
lag = []
for k in range(25):
    ctr.add(b + rank)
    lag.append(ctr.last_global)
import time
polled = 0
for _ in range(200):                       # poll(): non-blocking, folds in whatever has completed (here: the in-loop reductions)
    polled = ctr.poll()
    if polled >= 24 * (2 * b + 1):
        break
    time.sleep(0.01)
atotal = ctr.total()
tmax2 = sharding.reduce_max(1.0 + rank, torch.device("cpu"))
sharding.barrier()
print(json.dumps(dict(rank=rank, off=off, so=so, sb=sb, total=total, tmax=tmax, atotal=atotal, nred=ctr.reductions,
                      lag=lag[-1], polled=polled, tmax2=tmax2)), flush=True)
sharding.shutdown()
"""


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_two_rank_gloo_sharding_and_counter(tmp_path):
    import json
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert [d["off"] for d in outs] == [0, 32768]
    assert all(d["total"] == 10 * 32768 * 2 + 1 for d in outs)          # SUM over ranks
    assert all(d["tmax"] == 2.0 and d["tmax2"] == 2.0 for d in outs)    # MAX over ranks
    # asynchronous counter: exact total after the drain, 3 in-loop reductions + the final one; nothing is read back while
    # stepping (a read-back is a blocking device-to-host copy: it would drain the launch queue every `every` steps)
    assert all(d["atotal"] == 25 * (2 * 32768 + 1) and d["nred"] == 4 for d in outs)
    assert all(d["lag"] == 0 for d in outs)
    # poll() is the lagging running total: the third in-loop reduction (after 24 steps) once it has completed, never a blocking wait
    assert all(d["polled"] == 24 * (2 * 32768 + 1) for d in outs)
    # strong split covers the global batch exactly once, the boundary on a whole wavefront of the widest launch shape (64 envs; bit-for-bit
    # reproduction of the single-GPU run additionally needs the same launch shape: sharding.pin_global_shape, tests/test_gpu_api.py)
    assert outs[0]["so"] == 0 and outs[1]["so"] == outs[0]["sb"] and outs[0]["sb"] + outs[1]["sb"] == 32769
    assert outs[1]["so"] % 64 == 0


def test_shard_strong_partition():
    from random_envs_amd import sharding
    for world in (1, 2, 3, 4, 8):
        cover = []
        for r in range(world):
            off, b = sharding.shard_strong(32768 + 5, r, world)
            cover += list(range(off, off + b))
        assert cover == list(range(32768 + 5))
        # every boundary on a whole wavefront of the step kernels: sharded runs hold the single-GPU run's waves
        assert all(sharding.shard_strong(32768 + 5, r, world)[0] % sharding.WAVE_ENVS == 0 for r in range(world))
    assert sharding.shard_strong(40, 1, 4) == (10, 10)    # fewer whole waves than ranks: plain split


def test_shape_for_batch_restates_rex_create():
    """The host restatement of rex_create's launch-shape rule (what pin_global_shape pins a shard to); the GPU test
    test_launch_shape_follows_the_batch holds the library to the same table."""
    from random_envs_amd import sharding
    f = sharding.shape_for_batch
    assert f("hopper", 32768) == dict(lanes=64, pair=True, rolled=False, hum_pair=False)
    assert f("hopper", 16384) == dict(lanes=64, pair=True, rolled=False, hum_pair=False) and f("hopper", 8192)["lanes"] == 64
    assert f("walker2d", 16384)["lanes"] == 32 and f("walker2d", 8192)["lanes"] == 16 and f("halfcheetah", 8193)["lanes"] == 32
    assert f("walker2d", 8191)["lanes"] == 64 and f("halfcheetah", 1)["lanes"] == 64
    assert f("hopper", 32769) == dict(lanes=64, pair=False, rolled=False, hum_pair=False)
    assert f("hopper", 65537) == dict(lanes=64, pair=False, rolled=True, hum_pair=False)
    assert f("walker2d", 65537) == dict(lanes=64, pair=False, rolled=False, hum_pair=False)
    assert f("humanoid", 70000) == dict(lanes=64, pair=False, rolled=False, hum_pair=True)
    assert f("cartpole", 5) == dict(lanes=32, pair=False, rolled=False, hum_pair=False)
    assert f("halfcheetah", 4096, simds=64) == dict(lanes=64, pair=False, rolled=False, hum_pair=False)
    assert f("halfcheetah", 2048, simds=64) == dict(lanes=64, pair=True, rolled=False, hum_pair=False)
