import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, random_envs_amd as rex
for eid, n, amp in (("RandomHumanoid-v0", 3000, 0.4), ("RandomHopper-v0", 30000, 1.0), ("RandomWalker2d-v0", 10000, 1.0), ("RandomHalfCheetahNoisy-v0", 10000, 1.0)):
    B = 32768
    env = rex.make(eid, batch=B, seed=1)
    nom = torch.tensor(env.original_task)
    env.set_dr_distribution("uniform", torch.stack([0.8 * nom, 1.2 * nom], 1).flatten().tolist()); env.set_dr_training(True); env.reset()
    g = torch.Generator().manual_seed(0)
    acts = [(torch.rand(env.dims.act_dim, B, generator=g) * 2 * amp - amp).cuda() for _ in range(16)]
    t0 = time.time(); rs = 0.0; dn = 0
    for k in range(n):
        o, r, d = env.step_soa(acts[k % 16])[:3]
        if k % 500 == 0:
            rs += float(r.mean()); dn += int(d.sum()); assert torch.isfinite(o).all(), (eid, k)
    torch.cuda.synchronize()
    print(eid, "steps", n, "%.1f s" % (time.time() - t0), "counters", env.counters(), "mean r sample %.3f" % (rs / (n // 500 + 1)), flush=True)
    env.close()
