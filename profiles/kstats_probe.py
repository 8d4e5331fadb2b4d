import os, sys, ctypes
os.environ.setdefault("REX_LIB","librex_hip_kstats.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, random_envs_amd as rex
from random_envs_amd import _native
for eid in (sys.argv[1:] or ["RandomHopper-v0"]):
    B=32768
    env=rex.make(eid,batch=B,seed=0)
    nom=torch.tensor(env.original_task)
    env.set_dr_distribution("uniform", torch.stack([0.9*nom,1.1*nom],1).flatten().tolist()); env.set_dr_training(True); env.reset()
    g=torch.Generator().manual_seed(0)
    acts=[(torch.rand(env.dims.act_dim,B,generator=g)*2-1).cuda() for _ in range(8)]
    for k in range(100): env.step_soa(acts[k%8])
    torch.cuda.synchronize()
    out=(ctypes.c_ulonglong*8)(); _native.lib().rex_debug_kstats(out)
    for k in range(200): env.step_soa(acts[k%8])
    torch.cuda.synchronize()
    _native.lib().rex_debug_kstats(out); o=list(out)
    print(eid,'wave-solves',o[0],'pass1/solve %.2f pass2/solve %.2f ls_evals/solve %.2f nocon %.3f union-slots/solve %.2f fast-path %.3f'%(o[2]/o[0],o[3]/o[0],o[4]/o[0],o[5]/o[0],o[6]/o[0],o[7]/o[0]))
    env.close()
