"""Diagnostic: per-phase cycle shares of forward() from s_memtime stamps (build with -DREX_KTIME)."""
import os, sys, ctypes
os.environ["REX_LIB"] = "librex_hip_ktime.so"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, random_envs_amd as rex
from random_envs_amd import _native
for eid in sys.argv[1:] or ["RandomHopper-v0"]:
    B = 32768
    env = rex.make(eid, batch=B, seed=0)
    nom = torch.tensor(env.original_task)
    env.set_dr_distribution("uniform", torch.stack([0.9 * nom, 1.1 * nom], 1).flatten().tolist()); env.set_dr_training(True); env.reset()
    g = torch.Generator().manual_seed(0)
    acts = [(torch.rand(env.dims.act_dim, B, generator=g) * 2 - 1).cuda() for _ in range(8)]
    for k in range(100): env.step_soa(acts[k % 8])
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 96)()   # rex_debug_ktime copies all 96 accumulators
    _native.lib().rex_debug_ktime(out)   # read-and-zero: drops what the 100 warm-up steps accumulated
    for k in range(100): env.step_soa(acts[k % 8])
    torch.cuda.synchronize()
    _native.lib().rex_debug_ktime(out); o = list(out)
    n = o[7]; names = ["kinematics", "mass+bias+forces", "ldl(M)+qacc_smooth", "make_constraints", "solve"]
    tot = sum(o[:5])
    print(eid, "forward evals (waves):", n, "cycles/eval: " + ", ".join("%s %.0f (%.0f%%)" % (nm, o[i] / n, 100 * o[i] / tot) for i, nm in enumerate(names)),
          "| sum %.0f, whole step / 16 evals %.0f" % (tot / n, o[5] / n))
    env.close()
