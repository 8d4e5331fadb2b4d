#!/usr/bin/env python3
"""Record what a REAL MuJoCo does with the build's own model (run wherever `mujoco` or `mujoco_py` is installed; neither is
in the build image).  Writes tests/golden/live_mujoco_<kind>.json: the constants mj_setConst derives (body_invweight0,
dof_invweight0) and single env-step vectors (qpos, qvel, ctrl, xi) -> (qpos', qvel').  Commit the files: from then on
tests/test_live_mujoco.py::test_recorded_mujoco_vectors_if_present pins the oracle to them on every box, which moves the
MuJoCo-backed rows from "parity unpinned" to pinned.  Walker2d is recorded at its nominal
lengths (the lengths change the compiled model)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import live_mujoco as lm                                            # noqa: E402
from oracle_bindings import DIMS, oracle_constants                  # noqa: E402
from test_live_mujoco import _states                                # noqa: E402

if lm.have_mujoco() is None:
    sys.exit("no mujoco / mujoco_py importable: nothing recorded")
for kind in ("hopper", "walker2d", "halfcheetah", "humanoid"):
    n = 32 if kind == "humanoid" else 64
    q, v, a, xi = _states(kind, n, 77)
    if kind == "walker2d":
        from random_envs_amd.specs import SPECS
        xi[:, 7:11] = np.array(SPECS["walker2d"].nominal_task)[7:11]
    sim = lm.LiveSim(kind, oracle_constants(kind))
    qn, vn = [], []
    for i in range(n):
        sim.set_task(kind, xi[i])
        qq, vv = sim.step(q[i], v[i], a[i], DIMS[kind]["frame_skip"]); qn.append(qq.tolist()); vn.append(vv.tolist())
    rec = dict(mujoco=sim.version, body_invweight0=sim.compiled["body_invweight0"].tolist(), dof_invweight0=sim.compiled["dof_invweight0"].tolist(),
               qpos=q.tolist(), qvel=v.tolist(), ctrl=a.tolist(), xi=xi.tolist(), qpos_next=qn, qvel_next=vn)
    json.dump(rec, open(os.path.join(HERE, "live_mujoco_%s.json" % kind), "w"))
    print("recorded", kind, "with", sim.version)
