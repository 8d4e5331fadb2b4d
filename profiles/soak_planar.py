"""Soak: counters (solver_capped, nonfinite, overflow) after thousands of steps, episodic and endless.  `python soak_planar.py [B]`: the batch decides the
launch shape (32 768: two lanes per env; 65 536: one lane per env; 131 072: hopper on the rolled two-waves-per-SIMD kernel)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import random_envs_amd as rex
for eid, steps in [("RandomHopper-v0", 4000), ("RandomWalker2d-v0", 2500), ("RandomHalfCheetah-v0", 3000), ("RandomHopperUnmodeled-v0", 1000)]:
    for endless in (False, True):
        B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
        env = rex.make(eid, batch=B, seed=3)
        lo, hi = env.get_task_search_bounds()
        nom = torch.tensor(env.original_task)
        env.set_dr_distribution("uniform", torch.stack([0.7 * nom, 1.3 * nom], 1).flatten().tolist()); env.set_dr_training(True)
        env.set_endless(endless)
        env.reset()
        g = torch.Generator().manual_seed(0)
        acts = [((torch.rand(env.dims.act_dim, B, generator=g) * 2 - 1)).cuda() for _ in range(16)]
        n = (steps if not endless else steps // 2) * 32768 // max(B, 32768)
        for k in range(n):
            env.step_soa(acts[k % 16])
        torch.cuda.synchronize()
        c = env.counters()
        print(eid, "endless" if endless else "episodic", n, "steps x", B, "envs:", c, flush=True)
        env.close()
