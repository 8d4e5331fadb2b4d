"""Diagnostic: what the fused auto-reset costs a step launch (episodes that never end vs the normal mix)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, random_envs_amd as rex
for eid in sys.argv[1:] or ["RandomHopper-v0"]:
    for endless in (False, True):
        B = 32768
        env = rex.make(eid, batch=B, seed=0)
        nom = torch.tensor(env.original_task)
        env.set_dr_distribution("uniform", torch.stack([0.9 * nom, 1.1 * nom], 1).flatten().tolist()); env.set_dr_training(True)
        env.set_endless(endless); env.reset()
        g = torch.Generator().manual_seed(0)
        acts = [(torch.rand(env.dims.act_dim, B, generator=g) * 2 - 1).cuda() for _ in range(8)]
        for k in range(300): env.step_soa(acts[k % 8])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        nd = 0
        for k in range(1000):
            out = env.step_soa(acts[k % 8])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 1000
        print(eid, "endless" if endless else "episodic", "%.4f ms per step" % (dt * 1e3), "done fraction in the last step %.4f" % float(out[2].float().mean()), flush=True)
        env.close()
