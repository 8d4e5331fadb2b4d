"""Soak of the crowded-contact regime: humanoids that are never reset (endless, no time limit) end up lying and crumpled --
evaluations with more than 21 rows (the scratch-row PGS), every sweep size of the dual path.  Counters must stay 0."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, random_envs_amd as rex
B = 8192
env = rex.make("RandomHumanoid-v0", batch=B, seed=3, time_limit=False)
env.set_endless(True)   # random_env.py:51-60: done is never raised
nom = torch.tensor(env.original_task)
env.set_dr_distribution("uniform", torch.stack([0.8 * nom, 1.2 * nom], 1).flatten().tolist()); env.set_dr_training(True); env.reset()
g = torch.Generator().manual_seed(0)
acts = [(torch.rand(env.dims.act_dim, B, generator=g) * 0.8 - 0.4).cuda() for _ in range(16)]
t0 = time.time()
for k in range(600):
    o, r, d = env.step_soa(acts[k % 16])[:3]
    if k % 100 == 99:
        q, v = env.get_state(); torch.cuda.synchronize()
        print("step %d: %.1f s, z mean %.2f, finite %s, counters %s" % (k + 1, time.time() - t0, float(q[:, 2].mean()), bool(torch.isfinite(o).all()), env.counters()), flush=True)
env.close()
