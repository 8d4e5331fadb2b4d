"""Diagnostic: per-phase cycle shares of the humanoid forward() from s_memtime stamps (build with -DREX_KTIME)."""
import os, sys, ctypes
os.environ["REX_LIB"] = "librex_hip_ktime.so"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, random_envs_amd as rex
from random_envs_amd import _native
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
env = rex.make("RandomHumanoid-v0", batch=B, seed=0)
nom = torch.tensor(env.original_task)
env.set_dr_distribution("uniform", torch.stack([0.9 * nom, 1.1 * nom], 1).flatten().tolist()); env.set_dr_training(True); env.reset()
g = torch.Generator().manual_seed(0)
acts = [(torch.rand(env.dims.act_dim, B, generator=g) * 0.8 - 0.4).cuda() for _ in range(4)]
for k in range(60): env.step_soa(acts[k % 4])   # steady state: episodes end (z < 1) and restart all the time
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 96)(); _native.lib().rex_debug_ktime(out)
import time
t0 = time.perf_counter()
for k in range(8): env.step_soa(acts[k % 4])
torch.cuda.synchronize()
print("wall per step with stamps: %.2f ms" % ((time.perf_counter() - t0) / 8 * 1e3))
_native.lib().rex_debug_ktime(out); o = list(out)
n = o[17]
names = ["kinematics", "com_pos", "crb", "rne+forces", "collide", "make_constraints", "factor+solve", "pgs(MiJ+sweeps)"]
tot = sum(o[8:16])
print("forward evals with rows (waves):", n, "mean nefc(lane0) %.1f  mean sweeps %.1f" % (o[18] / n, o[19] / n))
for i, nm in enumerate(names): print("  %-18s %9.0f ticks/eval  %5.1f%%" % (nm, o[8 + i] / n, 100 * o[8 + i] / tot))
print("  pgs split: build A %.0f, sweeps %.0f, qacc %.0f" % (o[20] / n, o[21] / n, o[22] / n))
print("  evals where some lane of the wave took the scratch-row PGS (nefc > 21): %.1f%%" % (100.0 * o[23] / n))
h = o[24:96]; tot_h = sum(h)
print("  rows per evaluation (all lanes): " + " ".join("%d:%.2f%%" % (k, 100.0 * c / tot_h) for k, c in enumerate(h) if c))
print("  sum %.0f   stamped whole %.0f" % (tot / n, o[16] / n))
env.close()
