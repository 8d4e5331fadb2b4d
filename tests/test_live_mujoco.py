"""Live cross-check of the oracle (and, on a GPU, of the HIP path) against a stock MuJoCo when one is importable --
`mujoco` or `mujoco_py`, third-party packages, never the reference's code.  Skips cleanly where neither is installed
(this image); the MJCF emitter itself is tested everywhere.  See tests/live_mujoco.py and INTEGRATION.md section 4.

Reference boundary being checked: jinja_mujoco_env.py:94-95 (load_model_from_xml / MjSim) and :171-173 (sim.step)."""
import json
import os
import xml.etree.ElementTree as ET

import numpy as np
import pytest

import live_mujoco as lm
from oracle_bindings import DIMS, oracle_batch_step, oracle_constants, oracle_humanoid_step, rollout_states

KINDS = ["hopper", "walker2d", "halfcheetah", "humanoid"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs_mujoco = pytest.mark.skipif(lm.have_mujoco() is None, reason="no stock MuJoCo (mujoco / mujoco_py) importable here")


@pytest.mark.parametrize("kind", KINDS)
def test_emitted_mjcf_is_the_builds_own_model(kind):
    """well-formed MJCF with the tables' element counts and the oracle's inertials on every body"""
    c = oracle_constants(kind); t = lm.tables(kind)
    root = ET.fromstring(lm.emit_mjcf(kind, c))
    assert len(root.findall(".//body")) == len(t["bodies"]) == c["nbody"] - 1
    assert len(root.findall(".//joint")) == len(t["joints"]) and len(root.findall(".//geom")) == len(t["geoms"]) == c["ngeom"]
    assert len(root.findall(".//motor")) == len(t["motors"]) == c["nu"] and len(root.findall(".//pair")) == len(t["pairs"])
    masses = [float(e.get("mass")) for e in root.findall(".//inertial")]
    assert np.allclose(masses, c["body_mass"][1:], rtol=0, atol=0)
    assert root.find("compiler").get("inertiafromgeom") == "false"
    # nested bodies reproduce the world positions of the tables
    def walk(e, origin, out):
        for b in e.findall("body"):
            p = origin + np.array([float(x) for x in b.get("pos").split()]); out[b.get("name")] = p; walk(b, p, out)
    pos = {}; walk(root.find("worldbody"), np.zeros(3), pos)
    for b in t["bodies"]:
        assert np.allclose(pos[b["name"]], b["pos"], atol=1e-12)


def _states(kind, n, seed):
    d = DIMS[kind]; rng = np.random.RandomState(seed)
    if kind == "humanoid":
        from random_envs_amd.specs import SPECS
        nom = np.array(SPECS["humanoid"].nominal_task)
        q = np.tile(np.array([0, 0, 1.4, 1, 0, 0, 0] + [0] * 17, dtype=float), (n, 1)) + rng.uniform(-.01, .01, (n, 24))
        q[:, 7:] += rng.uniform(-.3, .3, (n, 17)); q[:, 2] = rng.uniform(1.0, 1.45, n)
        q[:, 3:7] /= np.linalg.norm(q[:, 3:7], axis=1, keepdims=True)
        v = rng.uniform(-1, 1, (n, 23)); xi = nom * rng.uniform(.8, 1.2, (n, 30)); a = rng.uniform(-.4, .4, (n, 17))
    else:
        q, v, xi = rollout_states(kind, n, steps_max=40, seed=seed); a = rng.uniform(-1, 1, (n, d["nu"]))
    return q, v, a, xi


def _oracle_step(kind, q, v, a, xi):
    return oracle_humanoid_step(q, v, a, xi) if kind == "humanoid" else oracle_batch_step(kind, q, v, a, xi, tolerance=0.0)


@needs_mujoco
@pytest.mark.parametrize("kind", KINDS)
def test_live_compile_constants(kind):
    """MuJoCo's own mj_setConst on the same inertials vs the oracle's restatement (diagApprox reads these)."""
    c = oracle_constants(kind); sim = lm.LiveSim(kind, c)
    assert np.allclose(sim.compiled["body_mass"], c["body_mass"], rtol=1e-12)
    assert np.allclose(sim.compiled["body_invweight0"], c["body_invweight0"], rtol=1e-6, atol=1e-9), kind
    assert np.allclose(sim.compiled["dof_invweight0"], c["dof_invweight0"], rtol=1e-6, atol=1e-9), kind


@needs_mujoco
@pytest.mark.parametrize("kind", KINDS)
def test_live_env_step_vs_oracle(kind):
    """frame_skip x mj_step of the real MuJoCo vs the fp64 oracle on identical (qpos, qvel, ctrl, xi)."""
    n = 64 if kind == "humanoid" else 200
    q, v, a, xi = _states(kind, n, 31); d = DIMS[kind]
    ref = _oracle_step(kind, q, v, a, xi)
    sim = lm.LiveSim(kind, oracle_constants(kind))
    worst = 0.0
    for i in range(n):
        if kind == "walker2d":      # lengths change the compiled model: one sim per env (random_walker2d.py:106-113)
            sim = lm.LiveSim(kind, oracle_constants(kind, size=xi[i, 7:11]))
        sim.set_task(kind, xi[i])
        qq, vv = sim.step(q[i], v[i], np.clip(a[i], -1, 1) if kind != "humanoid" else a[i], d["frame_skip"])
        worst = max(worst, np.abs(vv - ref["qvel"][i]).max() / (1 + np.abs(ref["qvel"][i]).max()), np.abs(qq - ref["qpos"][i]).max())
    print(kind, "live MuJoCo %s vs oracle: worst %.3e" % (sim.version, worst))
    assert worst < 1e-5, worst


@needs_mujoco
@pytest.mark.gpu
@pytest.mark.parametrize("kind,eid", [("hopper", "RandomHopper-v0"), ("halfcheetah", "RandomHalfCheetah-v0"), ("humanoid", "RandomHumanoid-v0")])
def test_live_env_step_vs_hip(kind, eid):
    """the HIP path through the C-ABI vs the real MuJoCo, at the stated fp32 tolerance, every lane"""
    import torch
    import random_envs_amd as rex
    n = 128; d = DIMS[kind]
    q, v, a, xi = [x.astype(np.float32).astype(np.float64) for x in _states(kind, n, 32)]
    env = rex.make(eid, batch=n, autoreset=False)
    env.set_task(xi.astype(np.float32)); env.set_state(q, v)
    env.step(torch.as_tensor(a, dtype=torch.float32))
    qq, vv = env.get_state(); qq = qq.cpu().numpy(); vv = vv.cpu().numpy()
    sim = lm.LiveSim(kind, oracle_constants(kind))
    for i in range(n):
        sim.set_task(kind, xi[i])
        ql, vl = sim.step(q[i], v[i], a[i], d["frame_skip"])
        assert np.abs(vv[i] - vl).max() / (1 + np.abs(vl).max()) < (5e-4 if kind == "humanoid" else 2e-4), (kind, i)
        assert np.abs(qq[i] - ql).max() < 2e-5 * (5 if kind == "humanoid" else 1), (kind, i)
    env.close()


@pytest.mark.parametrize("kind", KINDS)
def test_recorded_mujoco_vectors_if_present(kind):
    """tests/golden/dump_mujoco_vectors.py (run where a MuJoCo exists) records constants + single-step vectors into
    tests/golden/live_mujoco_<kind>.json; once committed, the oracle is pinned to them on every box."""
    path = os.path.join(ROOT, "tests", "golden", "live_mujoco_%s.json" % kind)
    if not os.path.exists(path):
        pytest.skip("no recorded MuJoCo vectors for %s (see INTEGRATION.md section 4)" % kind)
    rec = json.load(open(path))
    c = oracle_constants(kind)
    assert np.allclose(rec["body_invweight0"], c["body_invweight0"], rtol=1e-6, atol=1e-9)
    assert np.allclose(rec["dof_invweight0"], c["dof_invweight0"], rtol=1e-6, atol=1e-9)
    q, v, a, xi = [np.array(rec[k]) for k in ("qpos", "qvel", "ctrl", "xi")]
    ref = _oracle_step(kind, q, v, a, xi)
    assert np.abs(ref["qvel"] - np.array(rec["qvel_next"])).max() < 1e-5 and np.abs(ref["qpos"] - np.array(rec["qpos_next"])).max() < 1e-5
