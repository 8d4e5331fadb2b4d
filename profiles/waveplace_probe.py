"""Diagnostic: WHERE and WHEN the waves of one planar step launch ran (build with -DREX_WAVETIME; REX_LIB selects the library).
usage: waveplace_probe.py <env id> <batch>:<lanes> [<batch>:<lanes> ...]
Per launch shape: how many XCDs / CUs / SIMDs the workgroups landed on, the most waves any SIMD and any CU got, when waves started
relative to the first one, and how long they ran (100 MHz clock) -- grouped by how many waves shared their CU."""
import os, sys, ctypes, collections
os.environ["REX_ALLOW_TUNING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, random_envs_amd as rex
from random_envs_amd import _native

eid = sys.argv[1]
for spec in sys.argv[2:]:
    B, L = (int(x) for x in spec.split(":"))
    os.environ["REX_LANES"] = str(L)
    env = rex.make(eid, batch=B, seed=0)
    nom = torch.tensor(env.original_task)
    env.set_dr_distribution("uniform", torch.stack([0.9 * nom, 1.1 * nom], 1).flatten().tolist()); env.set_dr_training(True); env.reset()
    g = torch.Generator().manual_seed(0)
    acts = [(torch.rand(env.dims.act_dim, B, generator=g) * 2 - 1).cuda() for _ in range(8)]
    for k in range(300): env.step_soa(acts[k % 8])
    nb = (2 * B + L - 1) // L
    n = min(nb, 8192)
    env.enable_timing(True)
    info = (ctypes.c_ulonglong * (n * 8))(); torch.cuda.synchronize(); _native.lib().rex_debug_waveinfo(info, n)   # (zeroes the counters)
    st, du, late, per_simd, per_cu, by_share, mhz, phases = [], [], [], [], [], collections.defaultdict(list), [], []
    for k in range(10):
        env.step_soa(acts[k % 8]); torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * (n * 4))(); _native.lib().rex_debug_waveplace(buf, n)
        P = np.array(list(buf), dtype=np.uint64).reshape(n, 4)
        t0, t1, hw, xcc = P[:, 0].astype(np.int64), P[:, 1].astype(np.int64), P[:, 2].astype(np.int64), P[:, 3].astype(np.int64) & 15
        simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
        cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
        simd_key = cu_key * 4 + simd
        cs, cc = collections.Counter(simd_key.tolist()), collections.Counter(cu_key.tolist())
        per_simd.append(max(cs.values())); per_cu.append(max(cc.values()))
        first = t0.min()
        st.append((t0 - first) / 100.0); du.append((t1 - t0) / 100.0); late.append((t1.max() - first) / 100.0)
        ph = (ctypes.c_ulonglong * (n * 4))(); _native.lib().rex_debug_wavephase(ph, n)
        PH = np.array(list(ph), dtype=np.float64).reshape(n, 4)
        cyc = PH.sum(1)   # shader-clock cycles entry -> exit of the same waves
        phases.append(PH)
        mhz.append(cyc / ((t1 - t0) / 100.0))
        share = np.array([cc[c] for c in cu_key.tolist()])
        for s_ in np.unique(share): by_share[int(s_)].append(((t1 - t0)[share == s_] / 100.0))
        if k == 9:
            last = dict(xcds=len(set(xcc.tolist())), cus=len(cc), simds=len(cs), simd_hist=dict(sorted(collections.Counter(cs.values()).items())),
                        cu_hist=dict(sorted(collections.Counter(cc.values()).items())))
    ms = np.mean(env.read_timing()[-10:])
    S, D = np.concatenate(st), np.concatenate(du)
    print("%s batch %d, %d-lane blocks: %d blocks; kernel %.4f ms (events)" % (eid, B, L, nb, ms))
    print("   last launch: %(xcds)d XCDs, %(cus)d CUs, %(simds)d SIMDs; waves per SIMD -> SIMDs %(simd_hist)s; waves per CU -> CUs %(cu_hist)s" % last)
    print("   start after the first wave [us]: p50 %.2f  p90 %.2f  max %.2f | wave duration [us]: mean %.2f  p50 %.2f  p99 %.2f  max %.2f | first start -> last end %.2f us"
          % (np.percentile(S, 50), np.percentile(S, 90), max(s.max() for s in st), D.mean(), np.percentile(D, 50), np.percentile(D, 99), np.mean([d.max() for d in du]), np.mean(late)))
    _native.lib().rex_debug_waveinfo(info, n)
    I = np.array(list(info), dtype=np.float64).reshape(n, 8) / 10
    print("   per wave-step counts, mean over the waves: " + "  ".join("%s %.2f" % (nm, I[:, k].mean()) for k, nm in
          enumerate(["solves", "reset cycles", "pass1", "pass2", "ls_evals", "nocon", "slots_active", "selfpath"])))
    Mz = np.concatenate(mhz); PH = np.concatenate(phases)
    print("   cycles per wave [mean / max]: load state %.0f / %.0f | substeps %.0f / %.0f | reward, obs, stores %.0f / %.0f | fused reset %.0f / %.0f" %
          tuple(x for k in range(4) for x in (PH[:, k].mean(), PH[:, k].max())))
    print("   shader clock seen by the waves (cycles / duration): mean %.0f MHz  min %.0f  max %.0f; cycles per wave: mean %.0f" % (Mz.mean(), Mz.min(), Mz.max(), (Mz * D).mean()))
    print("   wave duration by waves sharing the CU: " + " | ".join("%d: mean %.2f max %.2f (n %d)" % (k, np.concatenate(v).mean(), np.concatenate(v).max(), len(np.concatenate(v)) // 10)
                                                                  for k, v in sorted(by_share.items())))
    env.close()
