"""Diagnostic: what the slowest waves of a humanoid step launch do differently (build with -DREX_KTIME -DREX_WAVETIME)."""
import os, sys, ctypes
os.environ["REX_LIB"] = "librex_hip_wavetime.so"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, random_envs_amd as rex
from random_envs_amd import _native
B = 32768
env = rex.make("RandomHumanoid-v0", batch=B, seed=0)
nom = torch.tensor(env.original_task)
env.set_dr_distribution("uniform", torch.stack([0.9 * nom, 1.1 * nom], 1).flatten().tolist()); env.set_dr_training(True); env.reset()
g = torch.Generator().manual_seed(0)
acts = [(torch.rand(env.dims.act_dim, B, generator=g) * 0.8 - 0.4).cuda() for _ in range(8)]
for k in range(80): env.step_soa(acts[k % 8])
rows = []
for k in range(12):
    env.step_soa(acts[k % 8]); torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * (1024 * 16))(); _native.lib().rex_debug_wavehum(out)
    rows.append(np.array(list(out), dtype=np.float64).reshape(1024, 16))
W = np.stack(rows)            # [launch, wave, slot]
names = ["smooth", "limit rows", "broad", "narrow loop", "  collide_pair", "  add_contact", "factor+solve", "build A", "sweeps", "qacc", "whole forward",
         "evals", "pair trips", "row trips", "sweeps (count)", "rows (sum)"]
tot = W[..., 10]
slow = tot > np.percentile(tot, 98); typ = tot < np.percentile(tot, 60)
print("whole-forward cycles per wave-step: mean %.0f  p90 %.0f  p99 %.0f  max %.0f -> max/mean %.2f" % (tot.mean(), np.percentile(tot, 90), np.percentile(tot, 99), tot.max(1).mean(), tot.max(1).mean() / tot.mean()))
print("%-18s %12s %12s %8s" % ("per wave-step", "typical(<p60)", "slow(>p98)", "ratio"))
for k, nm in enumerate(names): print("%-18s %12.0f %12.0f %8.2f" % (nm, W[..., k][typ].mean(), W[..., k][slow].mean(), W[..., k][slow].mean() / max(W[..., k][typ].mean(), 1)))
env.close()
