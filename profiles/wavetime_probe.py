"""Diagnostic: distribution over the waves of one launch of the cycles spent in the substeps (build with -DREX_WAVETIME).
At B = 32 768 every wave has its own SIMD, so the kernel time is the slowest wave's."""
import os, sys, ctypes
os.environ.setdefault("REX_LIB", "librex_hip_wavetime.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, random_envs_amd as rex
from random_envs_amd import _native
for eid in sys.argv[1:] or ["RandomHopper-v0"]:
    B = 32768
    env = rex.make(eid, batch=B, seed=0)
    nom = torch.tensor(env.original_task)
    env.set_dr_distribution("uniform", torch.stack([0.9 * nom, 1.1 * nom], 1).flatten().tolist()); env.set_dr_training(True); env.reset()
    g = torch.Generator().manual_seed(0)
    amp = 0.4 if "Humanoid" in eid else 1.0
    acts = [((torch.rand(env.dims.act_dim, B, generator=g) * 2 - 1) * amp).cuda() for _ in range(8)]
    for k in range(80 if "Humanoid" in eid else 300): env.step_soa(acts[k % 8])
    rows, infos = [], []
    info = (ctypes.c_ulonglong * (1024 * 8))(); torch.cuda.synchronize(); _native.lib().rex_debug_waveinfo(info, 1024)
    for k in range(20):
        env.step_soa(acts[k % 8]); torch.cuda.synchronize()
        out = (ctypes.c_ulonglong * 1024)(); _native.lib().rex_debug_wavetime(out, 1024)
        _native.lib().rex_debug_waveinfo(info, 1024)
        rows.append(np.array(list(out), dtype=np.float64)); infos.append(np.array(list(info), dtype=np.float64).reshape(1024, 8))
    w = np.stack(rows); I = np.stack(infos)
    if hasattr(_native.lib(), "rex_debug_wavephase") and "Humanoid" not in eid:
        ph = (ctypes.c_ulonglong * (1024 * 4))(); _native.lib().rex_debug_wavephase(ph, 1024)
        P = np.array(list(ph), dtype=np.float64).reshape(1024, 4)
        env.enable_timing(True); env.step_soa(acts[0]); torch.cuda.synchronize(); ms = env.read_timing()[-1]; env.enable_timing(False)
        print("   last launch, cycles per wave [mean / max]: load state %.0f / %.0f | substeps %.0f / %.0f | reward, obs, stores %.0f / %.0f | fused reset %.0f / %.0f | "
              "sum of means %.0f, slowest wave %.0f; kernel %.4f ms" % (P[:, 0].mean(), P[:, 0].max(), P[:, 1].mean(), P[:, 1].max(), P[:, 2].mean(), P[:, 2].max(),
                                                                     P[:, 3].mean(), P[:, 3].max(), P.sum(1).mean(), P.sum(1).max(), ms))
    if hasattr(_native.lib(), "rex_debug_evalphase") and "Humanoid" not in eid:
        ep = (ctypes.c_ulonglong * (1024 * 16))(); _native.lib().rex_debug_evalphase(ep, 1024)   # 16 slots per wave
        for k in range(10): env.step_soa(acts[k % 8])
        torch.cuda.synchronize(); _native.lib().rex_debug_evalphase(ep, 1024)
        E = np.array(list(ep), dtype=np.float64).reshape(1024, 16).mean(0) / 10
        nm = ["kinematics", "mass+bias+forces", "detect", "self / a0 / dispatch", "rows + solve", "  pass 1 (all)", "  pass 2 (all)"]
        print("   cycles per wave-step by phase of forward() (16 evaluations): " + " | ".join("%s %.0f" % (nm[k], E[k]) for k in range(7)) + " | sum of 0..4 %.0f" % E[:5].sum())
    print(eid, "cycles per wave-step: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f  -> max/mean %.2f" %
          (w.mean(), np.percentile(w, 50), np.percentile(w, 90), np.percentile(w, 99), w.max(1).mean(), w.max(1).mean() / w.mean()))
    print("   cycles in the fused reset path per wave-step: mean %.0f  p90 %.0f  max %.0f" % (I[..., 1].mean(), np.percentile(I[..., 1], 90), I[..., 1].max(1).mean()))
    print("   the same wave slow twice in a row (rank correlation of consecutive launches): %.2f" % np.corrcoef(w[:-1].ravel(), w[1:].ravel())[0, 1])
    names = ["solves", "reset cycles", "pass1", "pass2", "ls_evals", "nocon", "slots_active", "selfpath"]
    slow = w > np.percentile(w, 98); fast = w < np.percentile(w, 60)
    print("   per wave-step counts, typical waves (< p60) vs slow waves (> p98):")
    top = w.argmax(1)   # the slowest wave of each launch: what the launch waits for
    for k, nm in enumerate(names): print("     %-13s %8.1f %8.1f   slowest wave of each launch: mean %8.1f  min %8.1f  max %8.1f" %
                                         (nm, I[..., k][fast].mean(), I[..., k][slow].mean(), I[np.arange(len(top)), top, k].mean(), I[np.arange(len(top)), top, k].min(), I[np.arange(len(top)), top, k].max()))
    print("   per wave-step counts, mean over ALL waves: " + "  ".join("%s %.2f" % (nm, I[..., k].mean()) for k, nm in enumerate(names)))
    srt = np.sort(w, 1)
    print("   per launch: slowest wave %.0f, 2nd %.0f, 5th %.0f, 10th %.0f, 20th %.0f, 50th %.0f cycles (means over the launches)" %
          tuple(srt[:, -k].mean() for k in (1, 2, 5, 10, 20, 50)))
    env.close()
