import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np, random_envs_amd as rex
B = 32768
env = rex.make("RandomHumanoid-v0", batch=B, seed=0)
nom = torch.tensor(env.original_task)
env.set_dr_distribution("uniform", torch.stack([0.9 * nom, 1.1 * nom], 1).flatten().tolist()); env.set_dr_training(True); env.reset()
g = torch.Generator().manual_seed(0)
for k in range(80):
    env.step_soa((torch.rand(env.dims.act_dim, B, generator=g) * 0.8 - 0.4).cuda())
q, v = env.get_state()
np.savez("gpurun_out/hum_states.npz", q=q.cpu().numpy()[:4096], v=v.cpu().numpy()[:4096])
print(q.shape)
