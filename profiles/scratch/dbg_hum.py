import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import random_envs_amd as rex
from oracle_bindings import oracle_humanoid_step
from test_gpu_humanoid import _states
n = 1024
q, v, a, xi = _states(n, 3)
env = rex.make("RandomHumanoid-v0", batch=n, autoreset=False)
env.set_task(xi.astype(np.float32)); env.set_state(q, v)
obs, r, d, _ = env.step(torch.as_tensor(a, dtype=torch.float32))
ref = oracle_humanoid_step(q, v, a, xi)
qq, vv = env.get_state()
ev = np.abs(vv.cpu().numpy() - ref["qvel"]).max(1) / (1 + np.abs(ref["qvel"]).max(1))
bad = ev > 1e-3
print("bad lanes:", bad.sum(), "of", n)
print("bad by lane%32:", np.bincount(np.arange(n)[bad] % 32, minlength=32))
print("bad by block:", np.bincount(np.arange(n)[bad] // 32, minlength=32))
print("z of bad", np.round(q[bad, 2][:20], 2)); print("z of good", np.round(q[~bad, 2][:20], 2))
print("nan lanes:", np.isnan(ev).sum())
