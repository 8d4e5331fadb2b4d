"""Diagnostic: host-side cost of one rex_step call (enqueue only) per env kind, with / without auto-reset."""
import sys, time, ctypes, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, random_envs_amd as rex
from random_envs_amd import _native
L = _native.lib()
for eid in ["RandomCartPole-v0", "RandomHopper-v0", "RandomHumanoid-v0"]:
    for autoreset in (True, False):
        B = 64
        env = rex.make(eid, batch=B, seed=0, autoreset=autoreset)
        env.reset()
        a = torch.zeros(env.dims.act_dim, B, device='cuda', dtype=torch.int32 if env.dims.discrete_action else torch.float32)
        h = env._h; ap = ctypes.c_void_p(a.data_ptr()); op = ctypes.c_void_p(env._obs.data_ptr()); rp = ctypes.c_void_p(env._reward.data_ptr())
        dp = ctypes.c_void_p(env._done.data_ptr()); st = env._stream()
        for _ in range(20): L.rex_step(h, ap, op, rp, dp, None, None, st)
        torch.cuda.synchronize(); n = 500; t = time.perf_counter()
        for _ in range(n): L.rex_step(h, ap, op, rp, dp, None, None, st)
        t1 = time.perf_counter() - t; torch.cuda.synchronize(); t2 = time.perf_counter() - t
        print(eid, 'autoreset', autoreset, 'enqueue us/call %.1f  total us/call %.1f' % (t1 / n * 1e6, t2 / n * 1e6))
        env.close()
