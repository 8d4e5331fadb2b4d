"""Diagnostic: where a humanoid forward evaluation spends its time (build with -DREX_KTIME: s_memtime deltas summed in
registers per lane, wave maximum flushed once per kernel).  Steady-state batch: episodes end (z < 1) and restart all the time."""
import os, sys, ctypes, time
os.environ.setdefault("REX_LIB", "librex_hip_ktime.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, random_envs_amd as rex
from random_envs_amd import _native
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
env = rex.make("RandomHumanoid-v0", batch=B, seed=0)
nom = torch.tensor(env.original_task)
env.set_dr_distribution("uniform", torch.stack([0.9 * nom, 1.1 * nom], 1).flatten().tolist()); env.set_dr_training(True); env.reset()
g = torch.Generator().manual_seed(0)
acts = [(torch.rand(env.dims.act_dim, B, generator=g) * 0.8 - 0.4).cuda() for _ in range(4)]
for k in range(60): env.step_soa(acts[k % 4])
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 96)(); _native.lib().rex_debug_ktime(out)
t0 = time.perf_counter()
for k in range(8): env.step_soa(acts[k % 4])
torch.cuda.synchronize()
print("wall per step with stamps: %.2f ms" % ((time.perf_counter() - t0) / 8 * 1e3))
_native.lib().rex_debug_ktime(out); o = list(out)[8:24]
names = ["kinematics+com+rne+forces", "limit rows", "broad phase", "narrow loop (incl. pair + rows)", "  inside collide_pair", "  inside add_contact",
         "crb+factor+solve", "pgs: build A", "pgs: sweeps", "pgs: qacc", "whole forward"]
n = o[11]
print("wave-evaluations: %d; per evaluation: %.1f collide_pair calls, %.1f add_contact calls, %.1f sweeps (wave max), rows (wave max) %.1f" %
      (n, o[12] / n, o[13] / n, o[14] / n, o[15] / n))
for i, nm in enumerate(names): print("  %-34s %9.0f ticks/eval  %5.1f%%" % (nm, o[i] / n, 100.0 * o[i] / o[10]))
lv = list(out)[24:]
tot = sum(lv[8:16]) or 1
print("sweep levels (rows <= 4, 8, 10, 12, 14, 16, 18, 21): share of wave-evaluations with rows / cycles per evaluation in the sweeps / in the A build / share of all sweep cycles")
for k, nc in enumerate((4, 8, 10, 12, 14, 16, 18, 21)):
    if lv[k]: print("  NC %2d: %5.1f%%  %8.0f  %8.0f  %5.1f%%" % (nc, 100.0 * lv[k] / sum(lv[0:8]), lv[16 + k] / lv[k], lv[32 + k] / lv[k], 100.0 * lv[16 + k] / sum(lv[16:24])))
print("wave-evaluations by largest row count 1..21:", " ".join("%d:%.1f%%" % (k, 100.0 * lv[48 + k] / max(sum(lv[48:72]), 1)) for k in range(1, 22)))
env.close()
