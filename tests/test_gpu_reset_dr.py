"""reset()-time xi sampling (rocRAND Philox on device) and reset/auto-reset semantics.
RNG bits cannot match numpy: parity is on the distribution each dr_type defines
(random_env.py:148-203) and on the deterministic transforms around it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


def _ks_uniform(x, lo, hi):
    u = np.sort((x - lo) / (hi - lo)); n = len(u)
    return np.max(np.abs(u - (np.arange(n) + 0.5) / n))


def test_uniform_and_init_noise_hopper(torch_mod):
    import random_envs_amd as rex
    B = 65536
    env = rex.make("RandomHopper-v0", batch=B, seed=5)
    env.set_dr_distribution("uniform", [0.9, 1.1, 1.9, 2.1, 2.9, 3.1, 3.9, 4.1])      # README.md:58
    env.set_dr_training(True)
    obs = env.reset().cpu().numpy()
    xi = env.get_task().cpu().numpy()
    for k, (lo, hi) in enumerate([(0.9, 1.1), (1.9, 2.1), (2.9, 3.1), (3.9, 4.1)]):
        assert xi[:, k].min() >= lo and xi[:, k].max() <= hi and _ks_uniform(xi[:, k], lo, hi) < 0.01
    assert abs(np.corrcoef(xi[:, 0], xi[:, 1])[0, 1]) < 0.02
    q, v = env.get_state(); q = q.cpu().numpy(); v = v.cpu().numpy()
    q0 = np.array([0, 1.25, 0, 0, 0, 0])
    assert np.abs(q - q0).max() <= 0.005 + 1e-6 and np.abs(v).max() <= 0.005 + 1e-7          # random_hopper.py:113-114
    assert _ks_uniform((q - q0)[:, 3], -0.005, 0.005) < 0.01
    assert np.allclose(obs, np.concatenate([q[:, 1:], v], 1))
    # dr_training off: reset keeps the task
    env.set_dr_training(False); env.reset()
    assert np.array_equal(env.get_task().cpu().numpy(), xi)
    # without a distribution set_random_task raises like the reference (random_env.py:201)
    env2 = rex.make("RandomHopper-v0", batch=64)
    with pytest.raises(ValueError):
        env2.set_random_task()
    assert np.allclose(env2.get_task().cpu().numpy(), env2.original_task[None], rtol=1e-6)
    env.close(); env2.close()


def test_truncnorm_walker(torch_mod):
    """BASELINE config 4: truncnormal DR over the 13 Walker2d parameters."""
    import random_envs_amd as rex
    from random_envs_amd.specs import SPECS
    B = 32768
    spec = SPECS["walker2d"]; mean = np.array(spec.nominal_task); std = 0.1 * mean
    env = rex.make("RandomWalker2d-v0", batch=B, seed=2)
    env.set_dr_distribution("truncnorm", np.stack([mean, std], 1).ravel().tolist())
    env.set_dr_training(True); env.reset()
    xi = env.get_task().cpu().numpy().astype(np.float64)
    z = (xi - mean) / std
    assert np.abs(z).max() <= 2 + 1e-4                                    # a, b = -2, 2 (random_env.py:154)
    assert np.abs(z.mean(0)).max() < 0.03 and np.abs(z.std(0) - 0.8796).max() < 0.02    # std of N(0,1) cut at +-2
    # lower-bound rule: mean close to the bound -> two redraws then clamp (random_env.py:162-167)
    m2 = mean.copy(); m2[7] = 0.15; s2 = std.copy(); s2[7] = 0.1          # torsosize lower bound 0.1
    env.set_dr_distribution("truncnorm", np.stack([m2, s2], 1).ravel().tolist()); env.reset()
    t = env.get_task().cpu().numpy()[:, 7]
    assert t.min() >= 0.1 - 1e-7 and abs((t <= 0.1 + 1e-7).mean() - 0.2858 ** 3) < 0.01
    # the walker step still works on the re-derived per-env models
    obs, r, d, _ = env.step(torch_mod.zeros(B, 6))
    assert torch_mod.isfinite(obs).all()
    env.close()


def test_gaussian_and_fullgaussian(torch_mod):
    import random_envs_amd as rex
    B = 65536
    env = rex.make("RandomHalfCheetah-v0", batch=B, seed=4)
    mean = np.array([6, 1.5, 1.5, 1, 1.4, 1.2, 0.85, 0.4]); std = np.array([.5, .2, .2, .1, .1, .1, .1, 0.15])
    env.set_dr_distribution("gaussian", np.stack([mean, std], 1).ravel().tolist())
    env.set_random_task()
    xi = env.get_task().cpu().numpy().astype(np.float64)
    assert np.abs(xi[:, :7].mean(0) - mean[:7]).max() < 0.01 and np.abs(xi[:, :7].std(0) - std[:7]).max() < 0.01
    assert xi[:, 7].min() >= 0.1 - 1e-7                                   # redraw while < 0.1 (random_env.py:179-186)
    p = 0.02275  # P(N(0.4, 0.15) < 0.1)
    assert abs(env.counters()["gaussian_fail"] / B - p ** 3) < 1e-4       # the reference raises here; we clamp + count
    # fullgaussian: MVN in the normalised [0,4] space, clipped, denormalised (random_env.py:192-220)
    d = 8; m = np.full(d, 2.0); A = np.random.RandomState(0).randn(d, d) * 0.1; cov = A @ A.T + 0.05 * np.eye(d)
    env.set_dr_distribution("fullgaussian", {"mean": m, "cov": cov}); env.set_random_task()
    x = env.get_task().cpu().numpy().astype(np.float64)
    lo, hi = env.get_task_search_bounds()
    nrm = (x - lo) * 4 / (hi - lo)
    assert nrm.min() >= -1e-5 and nrm.max() <= 4 + 1e-5
    assert np.abs(nrm.mean(0) - m).max() < 0.01 and np.abs(np.cov(nrm.T) - cov).max() < 0.01
    env.close()


def test_autoreset_time_limit_and_noise(torch_mod):
    import random_envs_amd as rex
    torch = torch_mod
    B = 256
    env = rex.make("RandomHalfCheetahNoisy-v0", batch=B, seed=0)       # never "done": only TimeLimit ends episodes
    env.reset()
    for t in range(500):
        obs, r, d, info = env.step(torch.zeros(B, 6))
        if t < 499:
            assert not d.any()
    assert d.all() and info["TimeLimit.truncated"].all()                # max_episode_steps=500
    q, v = env.get_state()
    assert q.abs().max() <= 0.1 + 1e-6                                  # already reset (random_half_cheetah.py:124)
    # noisy obs: sigma = sqrt(1e-4) (random_half_cheetah.py:30,119)
    clean = torch.cat([q[:, 1:], v], 1)
    res = (obs - clean).cpu().numpy()
    assert abs(res.std() - 0.01) < 5e-4 and abs(res.mean()) < 5e-4
    env.close()
    # hopper auto-reset: done lanes restart from init noise with a fresh xi
    env = rex.make("RandomHopper-v0", batch=4096, seed=1)
    env.set_dr_distribution("uniform", [3.0, 4.0, 3.5, 4.5, 2.2, 3.2, 4.5, 5.5]); env.set_dr_training(True); env.reset()
    g = torch.Generator().manual_seed(0); seen = 0
    for t in range(60):
        xi_before = env.get_task().clone()
        obs, r, d, info = env.step(torch.rand(4096, 3, generator=g) * 2 - 1)
        if d.any():
            seen += int(d.sum())
            ob = obs[d].cpu().numpy()
            assert np.abs(ob[:, 0] - 1.25).max() <= 0.005 + 1e-6 and np.abs(ob[:, 5:]).max() <= 0.005 + 1e-6
            assert (env.get_task()[d] != xi_before[d]).any(1).all()      # fresh xi on reset
            assert torch.equal(env.get_task()[~d], xi_before[~d])
            term = info["terminal_observation"][d]
            assert ((term[:, 0] <= 0.7) | (term[:, 1].abs() >= 0.2) | (term.abs() >= 100).any(1)).all()
    assert seen > 1000
    env.close()
