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
for k in range(4): env.step_soa(acts[k % 4])
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 24)(); _native.lib().rex_debug_ktime(out)
for k in range(8): env.step_soa(acts[k % 4])
torch.cuda.synchronize()
_native.lib().rex_debug_ktime(out); o = list(out)
n = o[17]
names = ["kinematics", "com_pos", "crb", "rne+forces", "collide", "make_constraints", "factor+solve", "pgs(MiJ+sweeps)"]
tot = sum(o[8:16])
print("forward evals with rows (waves):", n, "mean nefc(lane0) %.1f  mean sweeps %.1f" % (o[18] / n, o[19] / n))
for i, nm in enumerate(names): print("  %-18s %9.0f ticks/eval  %5.1f%%" % (nm, o[8 + i] / n, 100 * o[8 + i] / tot))
print("  pgs split: build A %.0f, sweeps %.0f, qacc %.0f" % (o[20] / n, o[21] / n, o[22] / n))
print("  sum %.0f   stamped whole %.0f" % (tot / n, o[16] / n))
env.close()
