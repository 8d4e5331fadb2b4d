"""Diagnostic: cycles per phase of forward() and of the Newton iteration, per wave and env-step (build with -DREX_PHASES;
data-dependent s_memtime stamps, no other instrumentation).  Usage: REX_LIB=librex_hop_phases.so python profiles/phase_probe.py [env id]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, random_envs_amd as rex
from random_envs_amd import _native
eid = sys.argv[1] if len(sys.argv) > 1 else "RandomHopper-v0"
B = 32768
env = rex.make(eid, batch=B, seed=0)
nom = torch.tensor(env.original_task)
env.set_dr_distribution("uniform", torch.stack([0.9 * nom, 1.1 * nom], 1).flatten().tolist()); env.set_dr_training(True); env.reset()
g = torch.Generator().manual_seed(0)
acts = [((torch.rand(env.dims.act_dim, B, generator=g) * 2 - 1)).cuda() for _ in range(8)]
for k in range(300): env.step_soa(acts[k % 8])
ep = (ctypes.c_ulonglong * (1024 * 16))(); torch.cuda.synchronize(); _native.lib().rex_debug_evalphase(ep, 1024)
n = 20
for k in range(n): env.step_soa(acts[k % 8])
torch.cuda.synchronize(); _native.lib().rex_debug_evalphase(ep, 1024)
E = np.array(list(ep), dtype=np.float64).reshape(1024, 16).mean(0) / n
nm = ["kinematics", "mass+bias+forces", "detect", "self / a0 / dispatch", "rows + solve (all)", "pass 1", "pass 2 (all)", "  Hessian", "  factor + solve", "  phi'(1)", "  update", "  correction"]
print(eid, "cycles per wave and env-step:")
for k in range(12): print("  %-22s %9.0f" % (nm[k], E[k]))
print("  sum of the five forward() phases %.0f" % E[:5].sum())
