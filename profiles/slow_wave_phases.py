"""Diagnostic: what the SLOWEST wave of a launch spends its cycles on, phase by phase of forward(), against the mean wave
(-DREX_WAVETIME build WITHOUT -DREX_NOPHASES: s_memtime stamps around the phases, summed per wave and launch)."""
import os, sys, ctypes
os.environ.setdefault("REX_LIB", "librex_WTP_1.so")
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
torch.cuda.synchronize()
L = _native.lib()
nm = ["kinematics", "mass+bias+forces", "detect", "self/a0/dispatch", "rows+solve", "pass1", "pass2(all)", "H", "ldl+solve", "phi'", "update", "correction"]
top, mean, infos = [], [], []
ep = (ctypes.c_ulonglong * (1024 * 16))(); L.rex_debug_evalphase(ep, 1024)
info = (ctypes.c_ulonglong * (1024 * 8))(); L.rex_debug_waveinfo(info, 1024)
for k in range(40):
    env.step_soa(acts[k % 8]); torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 1024)(); L.rex_debug_wavetime(out, 1024)
    L.rex_debug_evalphase(ep, 1024); L.rex_debug_waveinfo(info, 1024)
    w = np.array(list(out), dtype=np.float64); E = np.array(list(ep), dtype=np.float64).reshape(1024, 16); I = np.array(list(info), dtype=np.float64).reshape(1024, 8)
    i = int(w.argmax()); top.append(np.concatenate([[w[i]], E[i, :12]])); mean.append(np.concatenate([[w.mean()], E[:, :12].mean(0)])); infos.append(I[i])
T = np.stack(top).mean(0); M = np.stack(mean).mean(0); I = np.stack(infos).mean(0)
print(eid, "cycles per wave-step: slowest wave of a launch %.0f, mean wave %.0f" % (T[0], M[0]))
for k, n in enumerate(nm): print("  %-18s slowest %8.0f   mean %8.0f   diff %8.0f" % (n, T[1 + k], M[1 + k], T[1 + k] - M[1 + k]))
print("  slowest wave counters: passes %.1f, selfpath evaluations %.1f" % (I[3], I[7]))
env.close()
